"""world_size-2 gloo test of the utterance sharding + single gather (the N > 1 path of bench.py), on CPU.
The per-rank synthesiser is the CPU oracle here (tests may use it); on GPUs it is LPCNetBatch."""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_everything_once():
    from dss_amd.distributed import shard_bounds
    for n in (0, 1, 7, 8, 255, 256, 8192):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.path.join({root!r}, "delayed-speech-synthesis_amd")); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import oracle_api
    from dss_amd.distributed import synthesize_sharded
    from dss_amd.lpcnet_weights import synthetic_blob, synthetic_features
    dist.init_process_group("gloo")
    orc = oracle_api.Oracle(os.path.join({root!r}, "oracle", "liboracle.so"))
    model = orc.lpcnet_model(synthetic_blob(0))
    N, F = 5, 4                                   # ragged: ranks get 3 and 2 utterances
    feats = np.stack([synthetic_features(300 + i, F) for i in range(N)])
    def synth(block):
        return torch.from_numpy(np.stack([orc.lpcnet_utterance(model, f) for f in block]).astype(np.int16))
    full = synthesize_sharded(feats, synth, dst=0)
    if dist.get_rank() == 0:
        want = np.stack([orc.lpcnet_utterance(model, f) for f in feats])
        assert full.shape == (N, F * 160) and np.array_equal(full.numpy(), want), "gathered PCM differs"
        print("GATHER_OK", flush=True)
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_rank_gloo_gather(tmp_path, oracle):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GATHER_OK" in out.stdout


def test_bench_multi_gpu_branch_dry_run():
    """bench.py's own N > 1 branch -- step() with distributed.gather_pcm, fence(), the MAX all-reduce over ranks and the JSON
    assembly -- run as two ranks under gloo with a stand-in synthesiser (DSS_BENCH_DRY=1), so that the first RCCL run on
    real GPUs executes only code that has run before (BASELINE config 4's sharded gather)."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", DSS_BENCH_DRY="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29613", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "6"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout                       # rank 0 prints ONE JSON line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    cfg = line["config"]
    assert cfg["world_size"] == 2 and cfg["batch_per_gpu"] == 6
    assert cfg["gathered_bytes_per_step"] == 2 * 6 * 100 * 160 * 2          # both shards' int16 PCM arrived on rank 0
    assert line["value"] > 0 and abs(line["samples_per_s_per_gpu"] * 2 - line["value"]) < 1e-6 * line["value"]
    assert "DRY RUN" in line["data"]                          # and says so: not a measurement
