"""prosper_amd: MI355X-native truncated-EM hot path behind prosper's CAModel plugin surface.

Only what the select_Hprimes -> E_step -> M_step path needs lives here:
  em/            EM driver, LinearAnnealing, CAModel and the device-backed models
  utils/         communicator (torch.distributed / RCCL), dlog sink, trace points
  csrc/          hand-written HIP kernels (gfx950) + the C ABI in include/prosper_hip.h
  _lib.py        ctypes binding of libprosper_hip.so
"""
__version__ = "0.1.0"
