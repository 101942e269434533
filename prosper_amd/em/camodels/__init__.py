"""Component-analysis model base: truncated state table + the E/M-step plugin surface.

Restates prosper/em/camodels/__init__.py: ``generate_state_matrix`` (:21-47) and
``CAModel`` (ctor :60-102, ``generate_data`` :104-122, ``select_partial_data`` :125-152,
``step`` :163-193, ``standard_init`` :196-235, ``compute_lpj`` :238-253).  Concrete
models supply ``select_Hprimes / E_step / M_step``; in this package those three run as
HIP kernels on the MI355X (see ``bsc_et``).
"""
from itertools import combinations

import numpy as np

from .. import Model
from ...utils import parallel
from ...utils import tracing
from ...utils.datalog import dlog


def generate_state_matrix(Hprime, gamma):
    """Binary Hprime-vectors with 2..gamma ones, ordered by ``itertools.combinations``
    for g = 2, 3, ..., gamma.  The row order is part of the ``logpj`` column contract.

    Returns ``(state_list, no_states, state_matrix (S, Hprime) uint8, state_abs (S,))``.
    """
    state_list = [np.array(s, dtype=np.int8)
                  for g in range(2, gamma + 1)
                  for s in combinations(range(Hprime), g)]
    no_states = len(state_list)
    state_matrix = np.zeros((no_states, Hprime), dtype=np.uint8)
    for i, s in enumerate(state_list):
        state_matrix[i, s] = 1
    state_abs = state_matrix.sum(axis=1)
    return state_list, no_states, state_matrix, state_abs


def _take_rows(val, sel):
    """Row subset for numpy arrays, torch tensors and device-array handles alike."""
    return val[sel]


class CAModel(Model):
    """Sparse-coding models with binary latents trained by Expectation Truncation."""

    def __init__(self, D, H, Hprime, gamma, to_learn=['W', 'pi', 'sigma'],
                 comm=parallel.COMM_WORLD):
        Model.__init__(self, comm)
        self.to_learn = to_learn
        self.D = D
        self.H = H
        self.Hprime = Hprime
        self.gamma = gamma

        assert Hprime <= H
        assert gamma <= Hprime

        tol = 1e-5
        self.noise_policy = {
            'W':     (-np.inf, +np.inf, False),
            'pi':    (tol, 1. - tol, False),
            'sigma': (0., +np.inf, False),
        }
        (self.state_list, self.no_states,
         self.state_matrix, self.state_abs) = generate_state_matrix(Hprime, gamma)

    # -- data ---------------------------------------------------------------
    def generate_data(self, model_params, my_N):
        """Draw ``my_N`` datapoints; RNG order as upstream (:119-120): one
        ``random((my_N, H))`` for the latents, then the model's own noise draw."""
        p = np.random.random(size=(my_N, self.H))
        s = p < model_params['pi']
        return self.generate_from_hidden(model_params, {'s': s})

    @tracing.traced
    def select_partial_data(self, anneal, my_data):
        """Random row subset of fraction ``anneal['partial']`` (0 and 1 mean all)."""
        partial = anneal['partial']
        if partial == 0 or partial == 1:
            return my_data
        my_N = my_data['y'].shape[0]
        my_pN = int(np.ceil(my_N * partial))
        if my_N == my_pN:
            return my_data
        sel = np.random.permutation(my_N)[:my_pN]
        sel.sort()
        # (per-row entries only: a cluster dict left behind by an earlier GSC.select_Hprimes describes other rows)
        return {key: _take_rows(val, sel) for key, val in my_data.items() if not isinstance(val, dict)}

    def check_params(self, model_params):
        return model_params

    # -- one EM step ------------------------------------------------------------
    @tracing.traced
    def step(self, anneal, model_params, my_data):
        """noisify -> check -> partial -> select_Hprimes -> E_step -> M_step -> log."""
        model_params = self.noisify_params(model_params, anneal)
        model_params = self.check_params(model_params)
        my_pdata = self.select_partial_data(anneal, my_data)
        my_pdata = self.select_Hprimes(model_params, my_pdata)
        my_joint_prob = self.E_step(anneal, model_params, my_pdata)
        new_model_params = self.M_step(anneal, model_params, my_joint_prob, my_pdata)
        dlog.append_all(new_model_params)
        dlog.append_all(anneal.as_dict())
        return new_model_params

    @tracing.traced
    def standard_init(self, data):
        """W = data mean + N(0, (sigma_init/4)^2) per column, sigma = mean per-dimension
        std, pi = 1/H (upstream :196-235).  Two collective means over the data."""
        comm = self.comm
        my_y = np.asarray(data['y'])
        my_N, D = my_y.shape
        assert D == self.D
        W_mean = parallel.allmean(my_y, axis=0, comm=comm)
        sigma_sq = parallel.allmean((my_y - W_mean) ** 2, axis=0, comm=comm)
        sigma_init = np.sqrt(sigma_sq).sum() / D
        W_init = W_mean[:, None] + np.random.normal(scale=sigma_init / 4., size=[D, self.H])
        return {'W': W_init, 'pi': 1. / self.H, 'sigma': sigma_init}

    def compute_lpj(self, anneal, model_params, my_data):
        """Candidates + log-pseudo-joints for ``my_data['y']`` (upstream :238-253)."""
        assert 'y' in my_data, "Key 'y' in my_data dict not defined."
        my_data = self.select_Hprimes(model_params, my_data)
        my_suff_stat = self.E_step(anneal, model_params, my_data)
        return my_suff_stat['logpj'], my_data['candidates']
