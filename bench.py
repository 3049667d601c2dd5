#!/usr/bin/env python3
"""bench.py -- headline benchmark: LPCNet 16 kHz synthesis throughput on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input: every rank resets its decoder
slots (a fresh LPCNet per utterance, as local/training.py:193 does), runs the frame-rate network and the
persistent sample-rate kernel over BATCH x 1-second utterances (100 x 20 float32 feature frames each,
already resident in HBM) and, for N > 1, the int16 PCM shards are collected on rank 0 with one RCCL gather
(the path's only exchange step).  Workload at N=1 = BASELINE.json configs[1]: batch 256, 1-s utterances.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "delayed-speech-synthesis_amd")
for p in (PKG, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

METRIC = "LPCNet 16kHz samples/s/GPU (×real-time) + ECoG→audio p50 latency"
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FRAMES = 100                   # 1-s utterances
FRAME = 160


# ---------------------------------------------------------------------------------------------------------
# cpu_baseline leg: the oracle (kind "port": this repo's scalar C restatement, gcc -O2 generic, which is how
# the reference's extension is effectively built) driven in the reference's pattern -- a process pool with one
# utterance per job and a fresh decoder per utterance (local/training.py:165-207).  Runs BEFORE the GPU is
# touched so that forking the pool is safe.
# ---------------------------------------------------------------------------------------------------------
def _cpu_job(args):
    seed, frames = args
    import oracle_api
    from dss_amd.lpcnet_weights import synthetic_blob, synthetic_features
    orc = oracle_api.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    m = orc.lpcnet_model(synthetic_blob(0))
    pcm = orc.lpcnet_utterance(m, synthetic_features(seed, frames))
    return int(pcm.shape[0])


def cpu_baseline(utt_per_core=2):
    import multiprocessing as mp
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "all"], stdout=subprocess.DEVNULL)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))              # GPU-box share for one GPU
    n_utts = cores * utt_per_core
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_job, [(10_000, 3)] * cores)                   # warm the workers (import, blob)
        t0 = time.time()
        samples = sum(pool.map(_cpu_job, [(s, FRAMES) for s in range(n_utts)], chunksize=1))
        dt = time.time() - t0
    return {"value": samples / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{n_utts} x 1-s utterances (seeds 0..{n_utts - 1}), one utterance per pool job, "
                      f"{cores}-process pool, oracle/liboracle.so (scalar C, gcc -O2 generic)",
            "x_realtime": samples / dt / 16000.0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="utterances per GPU (configs[1]: 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the streaming-latency leg (profiling runs)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        # convenience: re-launch under torch.distributed.run as a child (nothing has touched the GPU yet)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"), __file__,
               "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--batch", str(args.batch)] + (["--no-cpu-baseline"] if args.no_cpu_baseline else [])
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    import numpy as np
    import torch
    import torch.distributed as dist
    from dss_amd import _lib, lpcnet
    from dss_amd.lpcnet_weights import synthetic_features

    torch.cuda.set_device(local_rank)
    _lib.check(_lib.require_gpu().dss_set_device(local_rank))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    B = args.batch
    lpcnet.load_model(synthetic=True)           # seeded synthetic weights, explicitly (no checkpoint can be fetched offline)
    feats = torch.from_numpy(np.stack([synthetic_features(rank * B + b, FRAMES) for b in range(B)])).cuda()
    out = torch.empty((B, FRAMES * FRAME), dtype=torch.int16, device="cuda")
    wire = out.view(torch.uint8)                # RCCL has no int16 type: the PCM shard travels as bytes
    gathered = [torch.empty_like(wire) for _ in range(world)] if (world > 1 and rank == 0) else None
    dec = lpcnet.LPCNetBatch(B, FRAMES)

    def step():
        dec.reset_async()
        dec.synthesize_torch(feats, out=out)
        if world > 1:
            dist.gather(wire, gathered, dst=0)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # dominant kernel: live HIP-event timing of the sample-rate kernel on its launch stream, separate pass
    # (events add a host sync per step, so they are kept out of the timed region above)
    dec.enable_timing(True)
    for _ in range(max(3, min(args.steps, 10))):
        step()
    torch.cuda.synchronize()
    k_ms, f_ms = dec.kernel_ms(0), dec.kernel_ms(1)
    dec.enable_timing(False)

    # second half of BASELINE.json's metric: ECoG -> audio latency of the streaming mode (config 5), N=1 only
    latency = None
    if rank == 0 and world == 1 and not args.no_latency:
        from dss_amd.pipeline import StreamingPipeline
        sp = StreamingPipeline(128)
        sp.measure_latency(10)
        lat = sp.measure_latency(150)
        latency = {"p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)),
                   "config": "128 concurrent 64-ch ECoG streams, one 40-sample packet per stream per tick (4 frames): host "
                             "packet in -> HGA -> BiLSTM (chunk-wise, VAD gating off) -> LPCNet -> 640 int16 samples per "
                             "stream back on the host; structural floor of the reference (0.55 s + whole-segment "
                             "synthesis) not included"}

    # HBM traffic of the dominant kernel per launch: PMC counters cannot be read from inside this process, so the
    # figure comes from the committed rocprofv3 --pmc passes of this same command (profiles/*_pmc_traffic.json),
    # valid for the default workload only
    traffic = None
    try:
        if B == 256:
            with open(os.path.join(ROOT, "profiles", "r1i_pmc_traffic.json")) as f:
                traffic = float(json.load(f)["hbm_bytes_per_launch_corrected"]) / 1e9      # GB per launch
    except Exception:
        traffic = None

    if rank == 0:
        samples_per_step = world * B * FRAMES * FRAME
        value = samples_per_step * args.steps / dt
        bps = lpcnet.bytes_per_sample()                          # ~273 kB per output sample
        alg_bytes_per_launch = bps * B * FRAMES * FRAME
        achieved = alg_bytes_per_launch / (k_ms * 1e-3) / 1e9
        line = {
            "metric": METRIC, "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"LPCNet-only, batch={B} synthetic 1-s utterances per GPU (100x20 f32 features -> "
                                   f"16000 int16 samples each), fresh decoder state per utterance, features resident in HBM"
                                   + (", PCM shards gathered on rank 0 with one RCCL gather" if world > 1 else ""),
                       "batch_per_gpu": B, "frames": FRAMES, "weights": "synthetic seed 0 (xiph weights unobtainable offline)",
                       "parallelism": f"utterance-sharded x{world}"},
            "x_realtime": value / 16000.0, "samples_per_s_per_gpu": value / world,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "GB per launch (PMC, profiles/)",
                         "algorithmic_GB_per_launch": alg_bytes_per_launch / 1e9,
                         "kernel": "lpcnet_sample_kernel", "kernel_ms": k_ms, "frame_kernels_ms": f_ms,
                         "algorithmic_bytes_per_sample": bps,
                         "note": "algorithmic bytes = weights touched once per output sample at fp32 (SURVEY 8d); they are "
                                 "served from LDS/registers/L2, so frac > HBM share is expected; see profiles/ for PMC traffic"},
            "cpu_baseline": cpu,
            "latency": latency,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
