"""bench.py without a GPU: argument defaults, the self-start of the rank processes for
`python bench.py --gpus N`, the traffic key that names a kernel variant, and the sample ranges of
time shards."""
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_defaults_follow_the_configuration():
    a = bench.parse_args([])
    assert (a.gpus, a.config, a.frames, a.steps, a.warmup) == (1, 1, 430, 50, 25)
    a = bench.parse_args(["--config", "3", "--gpus", "8"])
    assert (a.frames, a.steps, a.warmup) == (430, 5, 2)
    a = bench.parse_args(["--config", "4", "--gpus", "8"])
    assert (a.frames, a.steps, a.warmup, a.transport) == (51680, 10, 3, "torch")
    a = bench.parse_args(["--gpus", "2", "--steps", "7", "--warmup", "1"])
    assert (a.steps, a.warmup) == (7, 1)


def test_plain_python_with_several_gpus_starts_the_ranks_itself(monkeypatch):
    """`python bench.py --gpus N` (no rank environment) must start N ranks under
    torch.distributed.run as a CHILD - before this process has touched the GPU - hand them its own
    arguments, and exit with their status."""
    import torch
    seen = {}

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7)

    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--config", "3", "--steps", "2"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "4", "--config", "3", "--steps", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert not torch.cuda.is_initialized()


def test_a_rank_does_not_start_ranks():
    a = bench.parse_args(["--gpus", "8"])
    assert bench.needs_self_launch(a, {})
    assert not bench.needs_self_launch(a, {"RANK": "3", "WORLD_SIZE": "8", "LOCAL_RANK": "3"})
    assert not bench.needs_self_launch(bench.parse_args([]), {})
    cmd = bench.launch_command(bench.parse_args(["--gpus", "2", "--master-port", "29999"]), ["--gpus", "2"])
    assert cmd[cmd.index("--master-port") + 1] == "29999"


def test_every_kernel_variant_has_a_traffic_key_of_its_own():
    def key(argv):
        a = bench.parse_args(argv)
        return bench.traffic_key(a.n_fft, bench.HOP, a.batch or 64, a.frames, a.lengths,
                                 bench.variant_name(a, a.frames_per_run))

    variants = [[], ["--unfused"], ["--inversion", "nnls"], ["--no-graph"], ["--traffic-only"], ["--alias-utts"],
                ["--linear", "0"], ["--recompute", "1"], ["--frames-per-run", "9"], ["--run-target", "1024"],
                ["--option", "deposit=0"], ["--single-frame", "1"], ["--config", "4"], ["--n-fft", "2048"],
                ["--unfused", "--no-graph"]]
    keys = [key(v) for v in variants]
    assert len(set(keys)) == len(keys)
    assert keys[0] == "n_fft1024_hop256_b64_t430_fixed_default"


@pytest.mark.parametrize("frames,world,n_fft", [(51680, 8, 1024), (51680, 3, 2048), (97, 4, 1024), (20, 1, 1024)])
def test_pieces_of_time_shards_cover_the_waveform(frames, world, n_fft):
    from tts_amd.distributed import split_frames
    r_cover = n_fft // bench.HOP
    ranges = split_frames(frames, world, min_frames=2 * (r_cover - 1))
    assert ranges[0][0] == 0 and ranges[-1][1] == frames
    assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    assert sum(bench.piece_samples(t0, t1, frames, r_cover) for t0, t1 in ranges) == bench.HOP * (frames - 1)
