"""The gated streaming leg of bench.py (row f4: the streaming mode AS decode_online.py RUNS IT), also runnable on its own:

    python tools/gated_leg.py [--lanes N] [--blocking]      ->  one JSON object

128 streams x 40-sample packets through HighGammaActivity -> FilterSpeechSegments (neural VAD on the library's kernel +
smoothing + segment ring) -> whole-segment BiLSTM -> LPCNet.  The tick never waits for a vocoder: closing segments go to side
streams (dss_amd/segment_queue.py) and come back by event.  Two passes over the same 10.4 s of input: UNPACED (ticks back to
back, then the queue drained: how much faster than the streams' own time the mode runs) and PACED at the amplifier's 40 ms
cadence with the host polling between ticks (segment-close -> PCM-on-host latency as a prosthesis would see it).
--blocking adds the reference's own behaviour (the tick waits for the segments it closed, units.py:531-538) for comparison."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "delayed-speech-synthesis_amd"),):
    if p not in sys.path:
        sys.path.insert(0, p)
FRAME = 160
S, TICKS, WARM = 128, 260, 10                                    # 10 s of stream time behind 10 warm-up ticks


def make_input():
    import numpy as np
    rng = np.random.default_rng(1)
    # loud / quiet stretches of 1 .. 4 s per stream
    env = np.empty((S, TICKS * 40))
    for s_ in range(S):
        t_, loud = 0, bool(rng.integers(2))
        while t_ < env.shape[1]:
            n_ = int(rng.integers(1000, 4000))
            env[s_, t_:t_ + n_] = 60.0 if loud else 3.0
            loud, t_ = not loud, t_ + n_
    # float32-valued, as an amplifier's packets are (local/units.py:78-82): the float64 rows and the wire-format payloads of the
    # two kinds of pass then carry the same numbers
    return [(rng.standard_normal((S, 40, 64)) * env[:, 40 * k:40 * k + 40, None]).astype(np.float32).astype(np.float64) for k in range(TICKS)]


def detector():
    """A seeded detector whose two logits mirror each other, so that its decision follows the input and both labels occur (no
    trained checkpoint exists offline)."""
    import torch
    from dss_amd.models import UnidirectionalVoiceActivityDetector
    torch.manual_seed(5)
    vad = UnidirectionalVoiceActivityDetector(nb_layer=2, nb_hidden_units=150, nb_electrodes=64)
    with torch.no_grad():
        vad.classifier.weight[1] = -vad.classifier.weight[0]
        vad.classifier.bias.zero_()
    return vad


def run(packets, paced, asynchronous=True, n_lanes=None, wire=False):
    import numpy as np
    from dss_amd.pipeline import GatedStreamingPipeline
    pct = lambda a, q: float(np.percentile(a, q)) if len(a) else None
    kw = {} if n_lanes is None else {"n_lanes": n_lanes}
    gp = GatedStreamingPipeline(S, 64, channel_means=np.full(64, 5.0), vad=detector(), max_segment_frames=1040,      # no segment can outgrow the 10.4 s of the leg
                                asynchronous=asynchronous, **kw)
    if wire:                                                     # the packets' bodies as they arrive: float32, channel-major
        packets = [np.ascontiguousarray(p.transpose(0, 2, 1), dtype=np.float32) for p in packets]
    push = gp.push_wire if wire else gp.push
    tick_ms, closing_ms, n_seg, seg_frames = [], [], 0, 0
    import gc
    gc.collect()
    gc.disable()                                                 # a generation-2 collection inside a tick is milliseconds; a real-time host pins this too
    t_start = time.perf_counter()
    t_meas = t_start
    for k in range(TICKS):
        if paced:                                                # the amplifier's cadence; the host polls while it waits:
            due = t_start + 0.04 * k                             # every 0.5 ms, and without sleeping over the last 2 ms (a host
            while True:                                          # that wakes from sleep INTO a tick pays ~0.25 ms of cold caches
                left = due - time.perf_counter()                 # and clocks on the tick; tools/gated_stages.py --paced shows it)
                if left <= 0:
                    break
                got = gp.poll()
                n_seg += len(got); seg_frames += sum(len(pcm) // FRAME for _, _, pcm in got)
                if left > 0.002:
                    time.sleep(0.0005)
        if k == WARM:
            t_meas = time.perf_counter()
            gp.queue.latencies_ms.clear()
            n_seg = seg_frames = 0
        closed_before = gp.segments_closed
        t0 = time.perf_counter()
        got = push(packets[k])
        ms = (time.perf_counter() - t0) * 1e3
        n_seg += len(got); seg_frames += sum(len(pcm) // FRAME for _, _, pcm in got)
        if k >= WARM:
            tick_ms.append(ms)
            if gp.segments_closed > closed_before:
                closing_ms.append(ms)
    t_ticks = time.perf_counter() - t_meas
    got = gp.flush()
    n_seg += len(got); seg_frames += sum(len(pcm) // FRAME for _, _, pcm in got)
    wall = time.perf_counter() - t_meas
    gc.enable()
    lat = list(gp.queue.latencies_ms)
    res = {"ticks": len(tick_ms), "tick_p50_ms": pct(tick_ms, 50), "tick_p99_ms": pct(tick_ms, 99), "tick_max_ms": max(tick_ms),
           "ticks_that_closed_a_segment": len(closing_ms), "closing_tick_p50_ms": pct(closing_ms, 50), "closing_tick_p99_ms": pct(closing_ms, 99),
           "segments": n_seg, "mean_segment_frames": (seg_frames / n_seg) if n_seg else None,
           "segment_close_to_pcm_on_host_p50_ms": pct(lat, 50), "segment_close_to_pcm_on_host_p99_ms": pct(lat, 99),
           "jobs": gp.queue.jobs_launched, "ticks_wall_s": t_ticks, "wall_s_incl_drain": wall,
           "vad_kernel": gp.vad_gpu is not None, "decoder_kernel": gp.dec_gpu is not None, "lanes": len(gp.queue.lanes)}
    del gp
    return res


def leg(n_lanes=None, blocking=False):
    packets = make_input()
    stream_s = (TICKS - WARM) * 0.04
    unpaced = run(packets, False, n_lanes=n_lanes)
    paced = run(packets, True, n_lanes=n_lanes)
    paced_wire = run(packets, True, n_lanes=n_lanes, wire=True)
    out = {"config": "128 streams x 40-sample packets through HGA -> VAD LSTM(150)x2 (csrc/vad_lstm.hip, one launch) -> gate kernel -> "
                     "event counts on the host; segments that close are collected (one launch on the tick's stream) and decoded + vocoded "
                     "on side streams (ragged csrc/bilstm_decoder.hip call + ragged LPCNet launch on a lane + PCM copy-out kernel per job; "
                     "dss_amd/segment_queue.py), push() returns the segments finished since the last tick; seeded detector, loud / quiet "
                     "synthetic input",
           "stream_seconds": stream_s, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
           "unpaced": unpaced, "unpaced_wall_over_stream_time": unpaced["wall_s_incl_drain"] / stream_s,
           "paced_40ms": paced, "paced_wall_over_stream_time": paced["wall_s_incl_drain"] / stream_s,
           "paced_40ms_wire_format_input": paced_wire,
           "note": "tick_* = wall time of push() over ALL measured ticks (host packet in -> event counts read, closing segments handed "
                   "to the queue, finished PCM handed back); segment_close_to_pcm = submit on the closing tick -> its PCM seen on the "
                   "host (a 3.46-s segment is 140 ms of vocoder time at single-utterance speed).  *_wire_format_input: the same pass with "
                   "push_wire(): each tick hands over the packets\' bodies as they arrive (float32, channel-major, 1.3 MB instead of 2.6 MB "
                   "of parsed float64 rows) and the reshape / transpose / float64 conversion runs on the device (dss_hga_extract_wire_dev).  BENCH_r04 (vocoding on the "
                   "tick path): closing ticks p50 145 ms, about 21 s of wall time for these 10 s of streams."}
    if blocking:
        b = run(packets, False, asynchronous=False)
        out["blocking_reference_behaviour"] = b
        out["blocking_wall_over_stream_time"] = b["wall_s_incl_drain"] / stream_s
    return out


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", type=int, default=None)
    ap.add_argument("--blocking", action="store_true")
    ap.add_argument("--hw-queues", type=int, default=8)
    a = ap.parse_args()
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(a.hw_queues))
    os.environ.setdefault("DSS_LPCNET_SYNTHETIC", "1")
    print(json.dumps(leg(a.lanes, a.blocking), indent=1))
