"""EM driver and model base class -- the drop-in boundary above the HIP kernels.

Restates prosper/em/__init__.py: ``Model`` (:24-110, incl. ``noisify_params`` :63-107 and
``gain`` :109-110) and ``EM`` (:115-178).  ``EM.run`` is the only entry point that
matters for the hot path: ``while not anneal.finished: model.step(...)``.
"""
import numpy as np

from ..utils import parallel
from ..utils import tracing
from ..utils.datalog import dlog


class Model(object):
    """Abstract EM model: data generation, E/M step, initialisation."""

    def __init__(self, comm=parallel.COMM_WORLD):
        self.comm = comm
        self.noise_policy = {}

    def generate_data(self, model_params, N):
        raise NotImplementedError

    def step(self, anneal, model_params, my_data):
        raise NotImplementedError

    def standard_init(self, data):
        raise NotImplementedError

    @tracing.traced
    def noisify_params(self, model_params, anneal):
        """Add annealed Gaussian noise to parameters named in ``self.noise_policy``
        (value = ``(low, high, absify)``), drawn on rank 0 and broadcast.

        Follows prosper/em/__init__.py:63-107 including its quirk that a *scalar*
        parameter is never actually clipped (the reference assigns the clipped value to
        an unused name, :85-88); array parameters are clipped to [low, high].
        """
        comm = self.comm
        normal = np.random.normal
        for param, (low, high, absify) in self.noise_policy.items():
            pvalue = model_params[param]
            scale = anneal[param + "_noise"]
            if scale != 0.0:
                if np.isscalar(pvalue):
                    new_pvalue = 0
                    if comm.rank == 0:
                        new_pvalue = pvalue + normal(scale=scale)
                        if absify:
                            new_pvalue = np.abs(new_pvalue)
                    pvalue = comm.bcast(new_pvalue)
                else:
                    if comm.rank == 0:
                        noisy = pvalue + normal(scale=scale, size=pvalue.shape)
                        noisy = np.minimum(high, np.maximum(low, noisy))
                        if absify:
                            noisy = np.abs(noisy)
                        pvalue = noisy
                    comm.Bcast([pvalue, parallel.DOUBLE])
            model_params[param] = pvalue
        return model_params

    def gain(self, old_params, new_params):
        return 0.


class EM(object):
    """Drives the annealed EM loop over ``model.step``."""

    def __init__(self, model=None, anneal=None, data=None, lparams=None, mpi_comm=None):
        self.model = model
        self.anneal = anneal
        self.data = data
        self.lparams = lparams
        self.mpi_comm = mpi_comm

    def step(self):
        """One EM step on the current parameters (result discarded, as upstream :138-149)."""
        self.model.step(self.anneal, self.lparams, self.data)

    def run(self, verbose=False):
        """Run a full cooling cycle (prosper/em/__init__.py:152-178)."""
        model, anneal, my_data = self.model, self.anneal, self.data
        model_params = self.lparams
        while not anneal.finished:
            if verbose:
                dlog.progress("EM step %d of %d" % (anneal['step'] + 1, anneal['max_step']),
                              anneal['position'])
            new_model_params = model.step(anneal, model_params, my_data)
            gain = model.gain(model_params, new_model_params)
            anneal.next(gain)
            if anneal.accept:
                model_params = new_model_params
            self.lparams = model_params
