"""Host logic of the look-ahead that lets an M-step launch the NEXT E-step (DeviceCAModel._predict_anneal): a schedule is a pure
function of its position (annealing.py:90-107), but only a predictor that was right about the current step is trusted, nothing
is predicted past the end, and a next step with parameter noise or partial data has inputs nobody knows yet.  No GPU: the
method reads the schedule and the model's noise policy only."""
import numpy as np

from prosper_amd.em.annealing import LinearAnnealing
from prosper_amd.em.camodels.bsc_et import BSC_ET


class _Plain(dict):
    def __missing__(self, k):
        return 0.0


def _reference_schedule(steps=50):
    an = LinearAnnealing(steps)
    an["T"] = [(0, 2.), (.7, 1.)]
    an["Ncut_factor"] = [(0, 0.), (2. / 3, 1.)]
    an["anneal_prior"] = False
    return an


def test_linear_schedule_is_read_one_position_ahead_once_the_predictor_has_been_right():
    m, an = BSC_ET(16, 8, 4, 3), _reference_schedule()
    track = []
    for it in range(50):
        nxt = m._predict_anneal(an)
        track.append(None if nxt is None else (nxt["T"], nxt["Ncut_factor"]))
        here = (an["T"], an["Ncut_factor"])
        an.next()
        if it + 1 < 50:
            ahead = (an["T"], an["Ncut_factor"])
            if it == 0:
                assert nxt is None                      # no history yet: nothing has shown that the caller advances the schedule
            else:
                assert track[-1] == ahead and ahead != here or ahead == here == track[-1], (it, track[-1], ahead)
        else:
            assert nxt is None                          # nothing is predicted past the schedule's end
    assert sum(t is not None for t in track) == 48
    assert an.finished


def test_a_schedule_that_is_not_advanced_gets_the_flat_predictor():
    m, an = BSC_ET(16, 8, 4, 3), _reference_schedule()
    an.next(); an.next()
    first = m._predict_anneal(an)
    assert first is None
    again = m._predict_anneal(an)                       # same position again: the caller does not advance -> same point predicted
    assert again is not None and (again["T"], again["Ncut_factor"]) == (an["T"], an["Ncut_factor"])
    an.next()                                           # now it moves -- to where the position-based look-ahead had pointed:
    ok = m._predict_anneal(an)                          # that predictor was right about THIS step and is trusted for the next
    pos = an.cur_pos
    assert ok is not None
    an.next()
    assert (ok["T"], ok["Ncut_factor"]) == (an["T"], an["Ncut_factor"]) and an.cur_pos == pos + 1
    # a caller that jumps (two positions at once): neither predictor was right, nothing is launched ahead
    m._predict_anneal(an)
    an.next(); an.next()
    assert m._predict_anneal(an) is None


def test_plain_mappings_and_noise():
    m = BSC_ET(16, 8, 4, 3)
    a = _Plain(T=1.3, Ncut_factor=0.4)
    assert m._predict_anneal(a) is None
    same = m._predict_anneal(_Plain(T=1.3, Ncut_factor=0.4))
    assert same is not None and same["T"] == 1.3          # a flat schedule handed in as fresh mappings
    assert m._predict_anneal(_Plain(T=1.2, Ncut_factor=0.4)) is None      # the temperature moved: not flat
    # parameter noise or partial data at the next point: its inputs are not known in advance
    an = LinearAnnealing(10)
    an["T"] = [(0, 1.), (1, 1.)]
    an["W_noise"] = [(0, 0.), (.45, 0.), (.55, 0.1), (1, 0.1)]
    seen = []
    for it in range(10):
        nxt = m._predict_anneal(an)
        seen.append(nxt is not None)
        an.next()
    assert seen[1] and seen[2] and not any(seen[5:]), seen
    assert np.isclose(an["W_noise"], 0.1)
