"""Annealing schedules for the EM driver.

Behavioural restatement of prosper/em/annealing.py: ``LinearAnnealing`` stores, per
named parameter, a list of ``(position, value)`` knots and interpolates linearly at the
current step (:90-107).  Quirks the drop-in keeps:
  * unknown names read as ``0.0`` (:93-94) -- the models rely on it for
    ``Ncut_factor``, ``anneal_prior``, ``partial``, ``*_noise``, ``data_noise``;
  * float positions are fractions of ``steps``, negative positions count from the end,
    a first knot is mirrored to position 0 and a last knot is held until ``steps + 1``
    (:63-88);
  * ``max_step``, ``position``, ``step`` are ordinary tracks (:56-60);
  * ``crit_params`` is an (always empty) list (:61).
"""
import numpy as np


class Annealing(object):
    """Base class: a cooling schedule plus loop control for ``EM.run``."""

    def reset(self):
        raise NotImplementedError

    def next(self, gain):
        raise NotImplementedError


class LinearAnnealing(Annealing):
    def __init__(self, steps=80):
        self.steps = steps
        self.anneal_params = {}
        self.reset()
        self['max_step'] = [(steps, steps)]
        self['position'] = [(0, 0.), (steps, 1.)]
        self['step'] = [(0, 0.), (steps, steps)]
        self.crit_params = []

    # -- schedule definition ------------------------------------------------
    def add_param(self, param_name, points):
        if np.isscalar(points):
            points = [(0, points)]
        knots = []
        for point in points:
            if not isinstance(point, tuple):
                raise TypeError("points must be a list of (pos, val)-tuples")
            pos, val = point
            if isinstance(pos, float):
                pos = int(pos * self.steps)
            if pos < 0:
                pos = self.steps + pos
            knots.append((pos, val))
        if knots[0][0] != 0:
            knots.insert(0, (0, knots[0][1]))
        if knots[-1][0] != self.steps:
            knots.append((self.steps + 1, knots[-1][1]))
        self.anneal_params[param_name] = knots

    def __setitem__(self, param_name, points):
        self.add_param(param_name, points)

    # -- lookup -----------------------------------------------------------------
    def __getitem__(self, param_name):
        knots = self.anneal_params.get(param_name)
        if knots is None:
            return 0.0
        cur = self.cur_pos
        # first knot strictly right of the current position; when none is, the last
        # segment is extrapolated (same arithmetic as the reference's loop-and-break)
        i = len(knots) - 1
        for j, (pos, _) in enumerate(knots):
            if pos > cur:
                i = j
                break
        left_pos, left_val = knots[i - 1]
        right_pos, right_val = knots[i]
        frac = float(cur - left_pos) / (right_pos - left_pos)
        return frac * (right_val - left_val) + left_val

    def as_dict(self):
        return {name: self[name] for name in self.anneal_params}

    # -- loop control -------------------------------------------------------------
    def reset(self):
        self.cur_pos = 0
        self.finished = False

    def next(self, gain=0.0):
        if self.finished:
            raise RuntimeError("Should not next() further when already finished!")
        self.accept = True
        self.cur_pos += 1
        if self.cur_pos >= self.steps:
            self.finished = True
