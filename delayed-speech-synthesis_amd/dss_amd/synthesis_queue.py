"""Offline bulk synthesis with the file contract of the reference's ``AsynchronousSynthesisQueue``
(local/training.py:165-207): every ``.npy`` file of LPCNet features (N x 20) becomes a 16 kHz ``.wav`` of the
same name.  The reference forks a pool and runs one utterance per process on a fresh ``LPCNet()``; here the
pending files are synthesised together, one persistent workgroup per utterance, in batched GPU launches.
(The rest of the reference's training.py -- dataset, checkpointing -- is training code and out of scope;
``python -m dss_amd.run train_bidirectional_model.py`` swaps this class into the user's own local.training.)"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import List

import numpy as np

logger = logging.getLogger("dss_amd.synthesis_queue")


class AsynchronousSynthesisQueue:
    MAX_BATCH = 1024        # files per launch (config 4's per-GPU share); 256 run at once, the rest queue behind them

    def __init__(self, nb_processes: int = 0):
        # nb_processes is accepted for signature compatibility; parallelism comes from the GPU batch
        self.nb_processes = nb_processes
        self._pending: List[str] = []

    def add_job(self, filename: str, verbose: int = 0):
        if verbose > 0:
            logger.info(f"Queued {filename} for synthesis.")
        self._pending.append(filename)
        if len(self._pending) >= self.MAX_BATCH:
            self._flush()

    def wait(self):
        """Synthesize everything still pending (the reference closes and joins its pool here)."""
        self._flush()

    @staticmethod
    def _generate_audio_from_lpc(lpc_filename: str, verbose: int = 0):
        q = AsynchronousSynthesisQueue()
        q.add_job(lpc_filename, verbose)
        q.wait()

    def _flush(self):
        from scipy.io.wavfile import write as wavwrite
        from .lpcnet import LPCNetBatch
        jobs, self._pending = self._pending, []
        loaded = []
        for name in jobs:
            try:    # like the reference, a bad file is logged and skipped, never fatal (training.py:196-198)
                feats = np.load(name).astype(np.float32)
                if feats.ndim != 2 or feats.shape[1] < 20 or feats.shape[0] == 0:
                    raise ValueError(f"expected (N, >=20) features, got {feats.shape}")
                loaded.append((name, np.ascontiguousarray(feats[:, :20])))
            except Exception as e:
                logger.error(f"Could not synthesize {name} due to an unexpected exceptions: {str(e)}")
        if not loaded:
            return
        # one ragged launch per MAX_BATCH files: every file gets a fresh decoder (training.py:193) and its own frame
        # count; rows are dispatched longest first so short files fill in behind long ones
        def emit(name, wav):
            try:
                wavwrite(Path(name).with_suffix(".wav").as_posix(), 16000, wav)
            except Exception as e:          # one unwritable path never stops the others (training.py:196-198)
                logger.error(f"Could not synthesize {name} due to an unexpected exceptions: {str(e)}")

        for a in range(0, len(loaded), self.MAX_BATCH):
            chunk = loaded[a:a + self.MAX_BATCH]
            try:
                frames = max(f.shape[0] for _, f in chunk)
                pcm = LPCNetBatch(len(chunk), frames).synthesize_ragged([f for _, f in chunk])
            except Exception as e:
                # the batched launch failed (allocation, device error): fall back to one launch per file so that
                # every file that can be synthesised still is, and each failure names its file like the reference does
                logger.error(f"Batched synthesis of {len(chunk)} files failed ({e}); retrying file by file.")
                for name, f in chunk:
                    try:
                        emit(name, LPCNetBatch(1, f.shape[0]).synthesize(f[None])[0])
                    except Exception as e1:
                        logger.error(f"Could not synthesize {name} due to an unexpected exceptions: {str(e1)}")
                continue
            for (name, _), wav in zip(chunk, pcm):
                emit(name, wav)
