#!/usr/bin/env python3
"""bench.py -- headline benchmark: LPCNet 16 kHz synthesis throughput on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input: every rank resets its decoder
slots (a fresh LPCNet per utterance, as local/training.py:193 does), runs the frame-rate network and the
persistent sample-rate kernel over BATCH x 1-second utterances (100 x 20 float32 feature frames each,
already resident in HBM) and, for N > 1, the int16 PCM shards are collected on rank 0 with one RCCL gather
(the path's only exchange step).  Workload at N=1 = BASELINE.json configs[1]: batch 256, 1-s utterances;
at N>1 = configs[3]: 1024 utterances per GPU (8192 on 8 GPUs), gathered on rank 0 (tools/scale.sh runs 1/2/4/8).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import subprocess
import sys
import time

# The gated streaming leg keeps up to 7 segment jobs in flight on side streams next to the tick's stream; ROCm maps streams
# onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share one run one after the other.  Must be set
# before the HIP runtime initialises; an explicit setting of the caller's is respected.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "delayed-speech-synthesis_amd")
for p in (PKG, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

METRIC = "LPCNet 16kHz samples/s/GPU (×real-time) + ECoG→audio p50 latency"
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_VECTOR_PEAK_TF = 157.3    # MI355X_MICROARCH.md: peak FP32 (vector), counts an FMA as 2 flops
# The path's parity contract forbids fused multiply-add (-ffp-contract=off: the reference's extension is a generic
# x86-64 build, separate multiply and add), so a MAC costs two VALU issues and the reachable vector peak is half.
FP32_NOFMA_PEAK_TF = FP32_VECTOR_PEAK_TF / 2
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


def _hga_job(args):
    seed, n = args
    import numpy as np
    import oracle_api
    from dss_amd.hga import reference_filters
    from dss_amd.synthetic import synthetic_ecog
    orc = oracle_api.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    hg, fh, zi_hg, zi_fh = reference_filters(1000)
    filt = {"sos_hg": hg, "sos_fh": fh, "zi_hg": zi_hg, "zi_fh": zi_fh}
    xs = [synthetic_ecog(seed + k, 1040, 64) for k in range(n)]
    t0 = time.time()
    for x in xs:                                   # a fresh extractor per trial (prepare_corpus.py:147-176)
        orc.extractor(filt, 64).extract(x)
    return n * 1.04, time.time() - t0


def _hga_ref_job(args):
    """The same trials through the REFERENCE's own code: oracle/_ref/hga_optimized*.so -- extensions/hga/hga_optimized.pyx compiled
    unmodified in the build container (oracle/Makefile ref; it travels with the snapshot) -- driven as HighGammaExtractor.extract_features
    drives it (local/units.py:145-161): scipy.signal.sosfilt x 2 with carried state, WarmStartFrameBuffer.insert, compute_log_power_features."""
    seed, n = args
    import numpy as np
    from scipy.signal import sosfilt
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
    import hga_optimized as ref
    from dss_amd.hga import reference_filters
    from dss_amd.synthetic import synthetic_ecog
    hg, fh, zi_hg, zi_fh = reference_filters(1000)
    xs = [synthetic_ecog(seed + k, 1040, 64) for k in range(n)]
    t0 = time.time()
    for x in xs:                                   # a fresh extractor per trial (prepare_corpus.py:147-176)
        fb = ref.WarmStartFrameBuffer(frame_length=0.05, frame_shift=0.01, fs=1000, nb_channels=64)
        s_hg = np.repeat(zi_hg, 64, axis=-1).reshape([zi_hg.shape[0], zi_hg.shape[1], -1])      # units.py:131-132
        s_fh = np.repeat(zi_fh, 64, axis=-1).reshape([zi_fh.shape[0], zi_fh.shape[1], -1])
        y, s_hg = sosfilt(hg, x, axis=0, zi=s_hg)
        y, s_fh = sosfilt(fh, y, axis=0, zi=s_fh)
        np.asarray(ref.compute_log_power_features(fb.insert(y), 1000, 0.05, 0.01))
    return n * 1.04, time.time() - t0


def _have_hga_ref():
    import glob
    return bool(glob.glob(os.path.join(ROOT, "oracle", "_ref", "hga_optimized*.so")))


def cpu_baseline(utt_per_core=2):
    """LPCNet and HGA on the GPU box's host cores: `cores`-process pool (the reference's pattern) and ONE core.
    Pool figures = total work / wall time of the pool.map call, for both legs (a figure built from the jobs' own spans moved
    6x when one job was descheduled); the spans are kept as a secondary field."""
    import multiprocessing as mp
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "all"], stdout=subprocess.DEVNULL)
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    cores = max(1, min(affinity, 16))           # GPU-box share for one GPU
    n_utts = cores * utt_per_core
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_job, [(10_000, 3)] * cores)                   # warm the workers (import, blob)
        t0 = time.time()
        samples = sum(pool.map(_cpu_job, [(s, FRAMES) for s in range(n_utts)], chunksize=1))
        dt = time.time() - t0
        # HGA: 1.04-s x 64-channel trials (config 3's unit), 8 per job, two jobs per worker
        pool.map(_hga_job, [(900, 1)] * cores)                      # warm (filter tables, first call)
        t0 = time.time()
        hga_res = pool.map(_hga_job, [(1000 + 8 * j, 8) for j in range(2 * cores)], chunksize=1)
        dth = time.time() - t0
        hga_ref = None
        if _have_hga_ref():                                         # the reference's OWN extension module, same trials
            try:
                pool.map(_hga_ref_job, [(900, 1)] * cores)
                t0 = time.time()
                ref_res = pool.map(_hga_ref_job, [(1000 + 8 * j, 8) for j in range(2 * cores)], chunksize=1)
                dtr = time.time() - t0
                sec_r, t_r = _hga_ref_job((2000, 8))
                hga_ref = {"value": sum(r[0] for r in ref_res) / dtr, "unit": "stream-seconds/s (64 ch @ 1 kHz)", "cores": cores, "kind": "reference",
                           "sample": f"the same {16 * cores} trials through oracle/_ref/hga_optimized (the reference's extensions/hga/hga_optimized.pyx, compiled "
                                     "unmodified) + scipy.signal.sosfilt x 2, driven as local/units.py:145-161 drives them; total stream-seconds / wall time of the pool.map call",
                           "one_core": {"value": sec_r / t_r, "unit": "stream-seconds/s", "cores": 1, "sample": "8 trials, one process"}}
            except Exception as e:                                  # e.g. another Python ABI than the module was built for
                hga_ref = {"error": f"{type(e).__name__}: {e}"}
    _cpu_job((10_001, 3))
    t0 = time.time()
    one = sum(_cpu_job((s, FRAMES)) for s in range(4))              # four 1-s utterances on ONE core (this process)
    dt1 = time.time() - t0
    sec1, t1 = _hga_job((2000, 8))
    hga_seconds = sum(r[0] for r in hga_res)
    host = {"os_cpu_count": os.cpu_count(), "affinity_cores": affinity, "pool_size": cores}
    return {"value": samples / dt, "unit": "samples/s", "cores": cores, "kind": "port", "host": host,
            "sample": f"{n_utts} x 1-s utterances (seeds 0..{n_utts - 1}), one utterance per pool job, "
                      f"{cores}-process pool, oracle/liboracle.so (scalar C, gcc -O2 generic); total samples / wall time of the pool.map call",
            "x_realtime": samples / dt / 16000.0,
            "one_core": {"value": one / dt1, "unit": "samples/s", "cores": 1, "x_realtime": one / dt1 / 16000.0,
                         "sample": "4 x 1-s utterances (seeds 0..3) in one process"},
            "hga": {"value": hga_seconds / dth, "unit": "stream-seconds/s (64 ch @ 1 kHz)", "cores": cores, "kind": "port",
                    "sample": f"{16 * cores} trials of 1.04 s x 64 ch, fresh filter state per trial, 8 per pool job, {cores}-process pool, "
                              "oracle/liboracle.so (DF2T cascade + frame buffer + log power; bit-equal to the reference's "
                              "scipy sosfilt + Cython chain); total stream-seconds / wall time of the pool.map call",
                    "sum_over_slowest_job_span": hga_seconds / max(r[1] for r in hga_res),
                    "reference": hga_ref,
                    "one_core": {"value": sec1 / t1, "unit": "stream-seconds/s", "cores": 1, "sample": "8 trials, one process"}}}


# ---------------------------------------------------------------------------------------------------------
# Dry run of the N > 1 branch (tests/test_cpu_distributed.py): DSS_BENCH_DRY=1 replaces the GPU, the library and RCCL by
# CPU tensors, a stand-in synthesiser and gloo -- everything else (step / fence / MAX over ranks / the JSON line) is the
# code the first multi-GPU run executes.
# ---------------------------------------------------------------------------------------------------------
DRY = os.environ.get("DSS_BENCH_DRY") == "1"


class _DryDecoder:
    """Stand-in for LPCNetBatch: deterministic int16 from the features, on the CPU."""

    def reset_async(self):
        pass

    def synthesize_torch(self, feats, out=None):
        import torch
        pcm = (feats[:, :, :1] * 1000.0).to(torch.int16).expand(-1, -1, FRAME).reshape(feats.shape[0], -1)
        if out is not None:
            out.copy_(pcm)
            return out
        return pcm

    def enable_timing(self, on):
        pass

    def kernel_ms(self, which):
        return 1.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None,
                    help="utterances per GPU (default: 256 = configs[1] at --gpus 1, 1024 = configs[3] at --gpus > 1)")
    ap.add_argument("--latency-ticks", type=int, default=1500, help="streaming-latency leg: 1500 ticks = 60 s of stream time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the streaming-latency leg (profiling runs)")
    args = ap.parse_args()

    if args.batch is None:
        args.batch = 256 if args.gpus == 1 else 1024
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
    if rank == 0 and not args.no_cpu_baseline and not DRY:      # before anything touches the GPU (the pool forks); N > 1: rank 0 too
        cpu = cpu_baseline()

    import numpy as np
    import torch
    import torch.distributed as dist
    from dss_amd.lpcnet_weights import synthetic_features

    B = args.batch
    if DRY:                                     # CPU tensors, gloo, a stand-in synthesiser: see _DryDecoder
        dev = "cpu"
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
        dec = _DryDecoder()
        info = {"fast_path": 1, "h_lds_bytes": 0, "kernel": "stand-in (DSS_BENCH_DRY=1)"}
        cus = 256

        def sync():
            pass
    else:
        from dss_amd import _lib, lpcnet
        dev = "cuda"
        torch.cuda.set_device(local_rank)
        _lib.check(_lib.require_gpu().dss_set_device(local_rank))
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        lpcnet.load_model(synthetic=True)       # seeded synthetic weights, explicitly (no checkpoint can be fetched offline)
        dec = lpcnet.LPCNetBatch(B, FRAMES, device=local_rank)
        info = lpcnet.model_info()
        cus = torch.cuda.get_device_properties(local_rank).multi_processor_count
        sync = torch.cuda.synchronize
    feats = torch.from_numpy(np.stack([synthetic_features(rank * B + b, FRAMES) for b in range(B)])).to(dev)
    out = torch.empty((B, FRAMES * FRAME), dtype=torch.int16, device=dev)

    def kernel_for(n_utts):     # the library's own rule (csrc/lpcnet_sample.hip dss_launch_sample_network)
        pair = info["fast_path"] == 1 and info["h_lds_bytes"] <= 144896 and n_utts > max(128, cus)
        return "lpcnet_sample_pair_kernel (two utterances per workgroup)" if pair else info["kernel"]
    from dss_amd.distributed import gather_pcm     # the collective the world-size-2 gloo test covers (tests/test_cpu_distributed.py)
    gathered_rows = [0]

    def step():
        dec.reset_async()
        dec.synthesize_torch(feats, out=out)
        if world > 1:
            full = gather_pcm(out, world * B, dst=0)            # int16 shards as bytes, one RCCL gather, rank order
            if full is not None:
                gathered_rows[0] = int(full.shape[0])

    def fence():
        if world > 1:
            dist.barrier()
        sync()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # dominant kernel: live HIP-event timing of the sample-rate kernel on its launch stream, separate pass
    # (events add a host sync per step, so they are kept out of the timed region above)
    dec.enable_timing(True)
    for _ in range(max(3, min(args.steps, 10))):
        step()
    sync()
    k_ms, f_ms = dec.kernel_ms(0), dec.kernel_ms(1)
    dec.enable_timing(False)

    extras = rank == 0 and world == 1 and not args.no_latency and not DRY      # the other configs' legs: N = 1 only
    # the kernel a model with skewed sparsity lands on: the generic sample-rate kernel, same batch, N=1 only
    generic = None
    if extras:
        dec.enable_trace(16)                    # development switch: force the generic kernel (no tracing)
        step(); torch.cuda.synchronize()
        tg = time.perf_counter()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        tg = (time.perf_counter() - tg) / 2
        dec.enable_trace(0)
        generic = {"kernel": "lpcnet_sample_generic_kernel", "ms_per_step": tg * 1e3, "value": B * FRAMES * FRAME / tg,
                   "unit": "samples/s", "note": "GRU A blocks streamed from L2 every sample: what a model runs on whose sparsity is "
                   "too skewed even for the CU-resident kernel's tail paths (over 16 z/r blocks of a row group beyond its register "
                   "slots, over 64 h blocks, or an LDS image over 151552 B); dss_lpcnet_model_info reports it, "
                   "profiles/r2_model_fit.txt has the steps in between"}

    # BASELINE.json configs[3], one GPU's share: 1024 utterances in one call (two utterances per workgroup beyond one per CU)
    config4 = None
    if extras:
        B4 = 1024
        f4 = torch.from_numpy(np.stack([synthetic_features(b, FRAMES) for b in range(B4)])).cuda()
        o4 = torch.empty((B4, FRAMES * FRAME), dtype=torch.int16, device="cuda")
        d4 = lpcnet.LPCNetBatch(B4, FRAMES, device=local_rank)

        def step4():
            d4.reset_async()
            d4.synthesize_torch(f4, out=o4)
        step4(); torch.cuda.synchronize()
        t4 = time.perf_counter()
        for _ in range(3):
            step4()
        torch.cuda.synchronize()
        t4 = (time.perf_counter() - t4) / 3
        d4.enable_timing(True)
        for _ in range(3):
            step4()
        torch.cuda.synchronize()
        k4 = d4.kernel_ms(0)
        d4.enable_timing(False)
        config4 = {"workload": "configs[3] per-GPU share: 1024 synthetic 1-s utterances, one call, fresh decoders, features in HBM",
                   "ms_per_step": t4 * 1e3, "value": B4 * FRAMES * FRAME / t4, "unit": "samples/s", "x_realtime": B4 * FRAMES * FRAME / t4 / 16000.0,
                   "kernel": kernel_for(B4),
                   "kernel_ms": k4, "kernel_ms_source": "HIP events on the launch stream"}
        del d4, f4, o4

    # The bulk callers' real shape (local/training.py:182-198): files of different lengths, one ragged call, rows in arrival order
    ragged = None
    if extras:
        Br = 1024
        counts = np.random.default_rng(0).integers(50, 301, Br)              # 0.5 .. 3 s
        fmax = int(counts.max())
        orr = torch.empty((Br, fmax * FRAME), dtype=torch.int16, device="cuda")
        dr = lpcnet.LPCNetBatch(Br, fmax, device=local_rank)
        audio = float(counts.sum()) * FRAME / 16000.0

        def time_ragged(fr):
            def stepr():
                dr.reset_async()
                dr.synthesize_ragged_torch(fr, counts, out=orr)
            stepr(); torch.cuda.synchronize()
            tr = time.perf_counter()
            for _ in range(2):
                stepr()
            torch.cuda.synchronize()
            return (time.perf_counter() - tr) / 2
        # every row its own utterance (seed = row): distinct features, RNG streams and embedding rows, as 1024 real files would be
        tr = time_ragged(torch.from_numpy(np.stack([synthetic_features(b, fmax) for b in range(Br)])).cuda())
        # rounds 1-4 stacked ONE feature matrix (seed 0) into all rows -- best-case L2 locality; kept once, for the comparison
        tr_same = time_ragged(torch.from_numpy(np.stack([synthetic_features(0, fmax)] * Br)).cuda())
        ragged = {"workload": "1024 synthetic utterances of 0.5-3 s (uniform), every row its own features (seed = row), fresh decoders, "
                              "one ragged call, rows in arrival order (the library dispatches them by decreasing length; two rows per "
                              "workgroup beyond one row per CU)",
                  "ms_per_step": tr * 1e3, "audio_seconds": audio, "value": audio * 16000.0 / tr, "unit": "samples/s",
                  "x_realtime": audio / tr,
                  "same_features_in_every_row": {"ms_per_step": tr_same * 1e3, "x_realtime": audio / tr_same,
                                                 "note": "what rounds 1-4 reported: one feature matrix stacked into all 1024 rows"}}
        del dr, orr

    # BASELINE.json configs[2]: 64 segments of 64-channel ECoG (1.04 s) -> HGA -> z-score -> BiLSTM -> LPCNet -> PCM
    config3 = None
    if extras:
        from dss_amd.pipeline import SegmentPipeline
        from dss_amd.synthetic import synthetic_ecog
        B3 = 64
        ecog = torch.from_numpy(np.stack([synthetic_ecog(1000 + b, 1040, 64) for b in range(B3)])).cuda()
        pipe = SegmentPipeline(B3)
        pipe(ecog); torch.cuda.synchronize()
        t3 = time.perf_counter()
        for _ in range(3):
            pcm3 = pipe(ecog)
        torch.cuda.synchronize()
        t3 = (time.perf_counter() - t3) / 3
        config3 = {"workload": "configs[2]: 64 segments x 1.04 s x 64-ch synthetic ECoG @1 kHz -> HGA (fused kernel) -> z-score -> "
                               "BiLSTM (csrc/bilstm_decoder.hip: three launches; seeded weights) -> LPCNet -> int16 PCM, fresh extractor and decoder per segment",
                   "ms_per_step": t3 * 1e3, "audio_seconds": float(pcm3.shape[0] * pcm3.shape[1] / 16000.0),
                   "x_realtime": float(pcm3.shape[0] * pcm3.shape[1] / 16000.0 / t3),
                   "note": "64 workgroups on 256 CUs: bounded by the single-utterance speed of the sample-rate kernel, not by HGA or the BiLSTM"}
        del pipe

    # second half of BASELINE.json's metric: ECoG -> audio latency of the streaming mode (config 5), N=1 only
    latency = None
    if extras:
        from dss_amd.pipeline import StreamingPipeline
        sp = StreamingPipeline(128)
        sp.measure_latency(10)
        lat = sp.measure_latency(args.latency_ticks)
        latency = {"p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)),
                   "ticks": int(args.latency_ticks), "stream_seconds": args.latency_ticks * 0.04,
                   "config": "128 concurrent 64-ch ECoG streams, one 40-sample packet per stream per tick (4 frames): host "
                             "packet in -> HGA -> BiLSTM (csrc/bilstm_decoder.hip, chunk-wise, VAD gating off) -> LPCNet -> 640 int16 samples per "
                             "stream back on the host (steady-state tick replayed from a captured HIP graph); structural floor of the reference (0.55 s + whole-segment "
                             "synthesis) not included"}

    # The streaming mode AS decode_online.py RUNS IT (row f4): HighGammaActivity -> FilterSpeechSegments (neural VAD on the library's
    # kernel + smoothing + segment ring) -> whole-segment BiLSTM -> LPCNet, 128 streams.  The tick never waits for a vocoder:
    # closing segments go to side streams (dss_amd/segment_queue.py) and come back by event.  Two passes over the same 10.4 s of
    # input: UNPACED (ticks back to back, then the queue drained: how much faster than the streams' own time the mode runs) and
    # PACED at the amplifier's 40 ms cadence, the host polling between ticks (segment-close -> PCM-on-host latency as a
    # prosthesis would see it).
    latency_gated = None
    if extras:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import gated_leg                                              # tools/gated_leg.py: the leg, shared with its stand-alone runner
        latency_gated = gated_leg.leg()

    # Level 1 of the drop-in: what an UNCHANGED decode_online.py pays per 10 ms frame -- LPCNet.LPCNet().synthesize()
    # through the xiph ABI (lpcnet_synthesize: one state, one frame, host in / host out; replayed from a per-state HIP graph)
    level1 = None
    if extras:
        import LPCNet as _L1
        net = _L1.LPCNet()
        f1 = synthetic_features(1, 660)
        for t_ in range(60):
            net.synthesize(f1[t_])
        l1 = []
        for t_ in range(60, 660):
            t0 = time.perf_counter()
            net.synthesize(f1[t_])
            l1.append((time.perf_counter() - t0) * 1e3)
        level1 = {"call": "LPCNet.LPCNet().synthesize(features[20]) -> int16[160] (extensions/lpcnet/LPCNet.pyx:30-40 surface, local/units.py:534-535 caller)",
                  "calls": len(l1), "p50_ms": float(np.percentile(l1, 50)), "p99_ms": float(np.percentile(l1, 99)),
                  "x_realtime": 10.0 / float(np.percentile(l1, 50))}
        del net

    # Row a11 on its own: the decoder kernels (csrc/bilstm_decoder.hip, three launches per call) next to the same weights as the
    # PyTorch-ROCm module (MIOpen's launch chain), device time per call for the two shapes the lines above contain
    decoder_ms = None
    if extras:
        from dss_amd.decoder import BiLstmDecoderGPU
        from dss_amd.models import BidirectionalSpeechSynthesisModel
        torch.manual_seed(0)
        mod = BidirectionalSpeechSynthesisModel(nb_layer=2, nb_hidden_units=100, nb_electrodes=64).eval().cuda()

        def _dev_ms(fn, n=30):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n
        decoder_ms = {"model": "BidirectionalSpeechSynthesisModel(2 layers x 100 hidden units, 64 inputs), seeded weights, zero state per call"}
        for S_, T_, tag in ((128, 4, "128_streams_x_4_frames"), (64, 104, "64_segments_x_104_frames")):
            kd = BiLstmDecoderGPU(S_, T_, mod)
            zd = torch.randn((S_, T_, 64), dtype=torch.float64, device="cuda")
            with torch.no_grad():
                tm = _dev_ms(lambda: mod(zd.to(torch.float32), mod.create_new_initial_state(batch_size=S_, device="cuda")))
            decoder_ms[tag] = {"kernels_ms": _dev_ms(lambda: kd(zd)), "pytorch_rocm_module_ms": tm}
            del kd

    # HBM traffic and issue counters of the dominant kernel per launch: PMC counters cannot be read from inside this
    # process, so they come from the committed rocprofv3 --pmc passes of this same command (profiles/), and are quoted
    # only for the workload they were taken on
    traffic, counters = None, None
    tag = {256: "r5_b256", 1024: "r5_b1024"}.get(B)
    try:
        with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")) as f:
            traffic = float(json.load(f)["hbm_bytes_per_launch_corrected"]) / 1e9          # GB per launch
    except Exception:
        traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_issue.json")) as f:
            counters = json.load(f)
    except Exception:
        counters = None

    if rank == 0:
        samples_per_step = world * B * FRAMES * FRAME
        value = samples_per_step * args.steps / dt
        bps = 273218.5 if DRY else lpcnet.bytes_per_sample()     # ~273 kB per output sample
        synth = B * (FRAMES - 2) * FRAME                         # samples the sample-rate kernel really computes per launch
        # SURVEY.md 8(d) "algorithmic flops": 2 per weight touched (sparse blocks, diagonal, GRU B, the 8 visited dual-FC
        # nodes) + the embedding adds; every one of them is a separate fp32 multiply or add (no FMA by contract)
        macs = (bps - 2.5) / 4.0 - 16 - 3 * 1152
        flops_per_sample = 2.0 * macs + 2 * 16 + 3 * 1152
        achieved_tf = flops_per_sample * synth / (k_ms * 1e-3) / 1e12
        alg_bytes_per_launch = bps * synth
        hbm_equiv = alg_bytes_per_launch / (k_ms * 1e-3) / 1e9
        roofline = {
            "bound": "valu",                 # fp32 vector ALU without FMA; the kernel's real ceiling
            "achieved": achieved_tf, "peak": FP32_NOFMA_PEAK_TF, "unit": "TFLOP/s", "frac": achieved_tf / FP32_NOFMA_PEAK_TF,
            "traffic": traffic, "traffic_unit": "GB of HBM per launch (PMC, profiles/)",
            "kernel": kernel_for(B), "kernel_ms": k_ms, "frame_kernels_ms": f_ms,
            "algorithmic_flops_per_sample": flops_per_sample, "samples_per_launch": synth,
            "frac_of_fma_peak": achieved_tf / FP32_VECTOR_PEAK_TF,
            "note": "useful fp32 operations (SURVEY 8d algorithmic flops) per second of the sample-rate kernel, against the "
                    "fp32 vector peak WITHOUT fused multiply-add (157.3/2 TFLOP/s): bit-exactness with the scalar C reference "
                    "forbids FMA and MFMA.  Weights are resident in VGPRs/LDS, so HBM is not the bound (see "
                    "roofline_hbm_equiv and traffic).",
        }
        if counters:
            c = counters.get("derived", {})
            roofline["valu_issue_utilisation"] = c.get("valu_issue_utilisation")
            roofline["counters_from"] = f"profiles/{tag}_pmc_issue.json @ git {counters.get('git_sha', 'unknown')}"
        hbm = {"bound": "hbm", "achieved": hbm_equiv, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_equiv / HBM_PEAK_GBS,
               "algorithmic_bytes_per_sample": bps, "algorithmic_GB_per_launch": alg_bytes_per_launch / 1e9,
               "hbm_traffic_frac_of_peak": (traffic / (k_ms * 1e-3) / HBM_PEAK_GBS) if traffic else None,
               "note": "SURVEY 8(d) convention: weights touched once per output sample at fp32.  NOT a physical HBM figure -- "
                       "those bytes are served from VGPRs/LDS, which is why it exceeds the HBM peak; the HBM traffic the "
                       "counters see is hbm_traffic_frac_of_peak of peak."}
        line = {
            "metric": METRIC, "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not DRY else "DRY RUN (DSS_BENCH_DRY=1): stand-in synthesiser on CPU, not a measurement",
            "config": {"workload": (f"configs[1]: LPCNet-only, batch={B}" if world == 1 else
                                    f"configs[3]: LPCNet batch={world * B} sharded over {world} GPUs, {B}")
                                   + " synthetic 1-s utterances per GPU (100x20 f32 features -> "
                                   "16000 int16 samples each), fresh decoder state per utterance, features resident in HBM"
                                   + (", PCM shards gathered on rank 0 with one RCCL gather" if world > 1 else ""),
                       "batch_per_gpu": B, "frames": FRAMES, "weights": "synthetic seed 0 (xiph weights unobtainable offline)",
                       "parallelism": f"utterance-sharded x{world}",
                       "scaling_note": "N = 1 is configs[1] (256 utterances); N > 1 lines are configs[3] (1024 utterances per GPU) and compare "
                                       "with this line's weak_scaling_ref (= config4_per_gpu.value), not with its value",
                       "world_size": (dist.get_world_size() if world > 1 else 1),
                       "gathered_bytes_per_step": (int(gathered_rows[0]) * FRAMES * FRAME * 2 if world > 1 else 0)},
            "x_realtime": value / 16000.0, "samples_per_s_per_gpu": value / world,
            # N = 1 runs configs[1] (256 utterances, one per CU); N > 1 runs configs[3] (1024 per GPU, two per workgroup), whose
            # one-GPU share is config4_per_gpu below: THAT is the figure an N > 1 line's per-GPU rate compares with
            "weak_scaling_ref": ({"value": config4["value"], "unit": "samples/s per GPU at 1024 utterances per GPU (config4_per_gpu)",
                                  "note": "scaling efficiency of an N > 1 line = value / (N * weak_scaling_ref.value), not value / (N * this line's value)"}
                                 if config4 else None),
            "roofline": roofline,
            "roofline_hbm_equiv": hbm,
            "generic_kernel": generic,
            "config4_per_gpu": config4,
            "ragged_1024": ragged,
            "config3": config3,
            "cpu_baseline": cpu,
            "latency": latency,
            "latency_gated": latency_gated,
            "level1": level1,
            "decoder": decoder_ms,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
