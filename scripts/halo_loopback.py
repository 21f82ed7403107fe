#!/usr/bin/env python3
"""Per-iteration cost of the halo swap at the long-form shard size (6 460 frames of a longer
utterance, neighbours on both sides), one process on one GPU: no swap at all (timing floor, wrong
result), the swap as four device copies (pack / unpack), and the swap issued by the library over
RCCL with the rank as its own neighbour (send / recv of 2 x (n_fft - hop) floats each way per
iteration, on the compute stream, no host wait).  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from tts_amd.vocoder import GriffinLimPlan, HaloChain
    per_rank = int(sys.argv[1]) if len(sys.argv) > 1 else 6460
    n_fft, hop, n_iter = 1024, 256, 60
    dev = torch.device("cuda:0")
    f = n_fft // 2 + 1
    s = torch.from_numpy((np.random.default_rng(1).standard_normal((per_rank, f)) ** 2).astype(np.float32)).to(dev)
    plan = GriffinLimPlan(n_fft, hop, n_fft, 80, per_rank, 1, dev)
    chain = HaloChain(HaloChain.unique_id(), 0, 1, dev)
    plan.halo_attach(chain)
    plan.set_option("halo_loopback", 1)
    if os.environ.get("GLV_FPR"):
        plan.set_option("frames_per_run", int(os.environ["GLV_FPR"]))

    def copies():
        a, b = plan.shard_halo_pack()
        plan.shard_halo_unpack(a, b)

    def timed(swap):
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            plan.shard_begin(s, per_rank, 3 * per_rank, seed=3)
            swap()
            for _ in range(n_iter):
                plan.shard_step(0.99)
                swap()
            plan.shard_end()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return round(1e3 * best / (n_iter + 1), 4)

    waves = {}

    def one_call(form):
        plan.set_option("shard_graph", form)
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            plan.shard_begin(s, per_rank, 3 * per_rank, seed=3)
            plan.shard_run(n_iter, 0.99)
            y, _ = plan.shard_end()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        waves[form] = (y.clone(), plan.shard_last_run_form())
        return round(1e3 * best / (n_iter + 1), 4)

    by_form = {"enqueued": one_call(0), "one_graph": one_call(1), "one_graph_edges_apart": one_call(2),
               "automatic": one_call(-1)}
    out = {"frames_per_rank": per_rank, "n_iter": n_iter,
           "ms_per_iteration_library_rccl_loopback_one_call": by_form,
           "forms_taken": [waves[f][1] for f in (0, 1, 2, -1)],
           "same_bits": bool(all(torch.equal(waves[0][0], waves[f][0]) for f in (1, 2, -1))),
           "ms_per_iteration_no_swap": timed(lambda: None),
           "ms_per_iteration_device_copies": timed(copies),
           "ms_per_iteration_library_rccl_loopback": timed(plan.shard_exchange)}
    plan.halo_attach(None)
    plan.close()
    chain.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
