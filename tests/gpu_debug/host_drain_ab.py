"""RECORD of a round-4 experiment (the two library knobs it drove were removed again): host-pointer sign at 2^20 from page-locked arrays.
(a) downloads queued by the submitting thread after it waited for the piece's compute, so the download stream is idle when the copy arrives (PLUME_HOST_DRAIN_SYNC=1):
    20.4 / 20.6 ms -- the copies are still blit kernels (__amd_rocclr_copyBuffer in the kernel trace, no DEVICE_TO_HOST record in the memory-copy trace);
(b) a piece's downloads held back until the NEXT piece enters k_sign_hmul (PLUME_HOST_DRAIN_LATE=1), because beside the next piece's memory-bound kernels they cost those
    kernels up to x1.8 (k_sign_gmul) in the trace: 19.2 -> 27.3 ms median (best 21.8): the slot is released later and the submitting thread stalls on it.
Kept: tests/gpu_debug/e2e_trace_sign.py + trace_timeline.py, which show the pipeline."""
