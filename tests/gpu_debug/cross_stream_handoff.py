"""Round 3: what does a cross-stream dependency cost on this stack?  (LABNOTES.md §6: why the sub-batch overlap lost.)
N dependent kernels in ONE stream against the same N kernels alternating between TWO streams with an event hand-off in between (record on one, wait on the other),
for tiny kernels (latency) and for kernels that write a large buffer (write-back between queues).  Prints microseconds per kernel / per hand-off."""
import json
import time

import torch


def run(n_elems, reps=200):
    dev = torch.device("cuda:0")
    x = torch.zeros(n_elems, dtype=torch.float32, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    evs = [torch.cuda.Event() for _ in range(reps)]

    def same():
        with torch.cuda.stream(s1):
            for _ in range(reps):
                x.add_(1.0)

    def pingpong():
        for k in range(reps):
            a, b = (s1, s2) if k % 2 == 0 else (s2, s1)
            with torch.cuda.stream(a):
                x.add_(1.0)
                evs[k].record(a)
            b.wait_event(evs[k])

    out = {}
    for name, fn in (("one_stream", same), ("two_streams_event_handoff", pingpong)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        out[name + "_us_per_kernel"] = round((time.perf_counter() - t0) / reps * 1e6, 2)
    out["handoff_extra_us"] = round(out["two_streams_event_handoff_us_per_kernel"] - out["one_stream_us_per_kernel"], 2)
    return out


if __name__ == "__main__":
    for n in (1, 1 << 20, 1 << 26):
        print(json.dumps({"elements_f32": n, "bytes_written_per_kernel": 4 * n, **run(n)}), flush=True)
