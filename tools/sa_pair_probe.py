#!/usr/bin/env python3
"""How many 32-row blocks sa3 / sa4 run per step with the item lists (geometry.hip: sa_pair_kernel) on the bench workload, and the wall time of
a K-step call with and without them (LSDM_SA_PAIR, read when the context is created).  GPU box only:  python tools/sa_pair_probe.py [--surface]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    import bench
    surface = "--surface" in sys.argv
    B, K = 64, 7
    dev = torch.device("cuda", 0)
    for mode in ("1", "0", "1", "0"):
        os.environ["LSDM_SA_PAIR"] = mode
        r = bench.Runner("bf16x3", B, 13, K + 3, 0, 0, dev, surface=surface)
        x0 = r.x.clone()
        ts = []
        for rep in range(6):
            r.x.copy_(x0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r.run(r.T - 1, 0, K, invariant=True)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        line = "LSDM_SA_PAIR=%s  %d-step call: min %.3f ms  median %.3f ms" % (mode, K, min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3)
        if mode == "1":
            hip = ctypes.CDLL("libamdhip64.so")
            for name, S in (("sa3_items", 64), ("sa4_items", 16)):
                ptr, n = r.ctx.workspace_tensor(name)
                cnt = (ctypes.c_int * K)()
                assert hip.hipMemcpy(cnt, ctypes.c_void_p(ptr), ctypes.c_size_t(4 * K), 2) == 0
                groups = B * 9 * S
                line += "\n   %s: blocks per step / groups = %s" % (name, " ".join("%.3f" % (c / groups) for c in cnt))
        print(line, flush=True)
        del r
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
