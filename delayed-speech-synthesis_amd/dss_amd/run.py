"""Run one of the reference's scripts UNCHANGED on the GPU path:

    python -m dss_amd.run decode_online.py config.ini --run test --overwrite | play -t raw -r 16000 ...
    python -m dss_amd.run train_bidirectional_model.py ...

What it does before handing control to the script (runpy, ``__main__``):
  1. puts the drop-in extension modules (LPCNet.py, hga_optimized.py) first on sys.path, so ``import LPCNet`` and
     ``from hga_optimized import ...`` (local/units.py:7,23; local/training.py:13) resolve to libdss_hip.so;
  2. imports the USER'S OWN ``local.units`` / ``local.training`` (the script's directory is on sys.path, as when it is
     run directly) and replaces in them exactly the classes whose work moves to the GPU:
        local.units.HighGammaExtractor      -> dss_amd.units.HighGammaExtractor      (fused IIR + framing + log power)
        local.units.DelayedLPCNetVocoder    -> dss_amd.units.DelayedLPCNetVocoder    (whole segment per launch)
        local.units.RecurrentNeuralDecodingModel -> dss_amd.units.gpu_decoding_unit(<the user's class>): a SUBCLASS of the
                                               user's own unit -- its initialize() is the user's, its decode() runs the
                                               reference's decoder architecture on the library's kernels (other modules
                                               go through the user's own handler)
        local.training.AsynchronousSynthesisQueue -> dss_amd.synthesis_queue.AsynchronousSynthesisQueue
     ``HighGammaActivity.initialize`` (units.py:199-201) looks HighGammaExtractor up in its module at run time, so the
     reference's own unit class picks the GPU extractor up; every other unit is the user's code, untouched.
Nothing of the reference is copied or shadowed; a module that cannot be imported (missing zmq/mne/ezmsg) is reported and
left alone (the extension swap of step 1 still applies).
"""
from __future__ import annotations

import importlib
import os
import runpy
import sys

DROPIN_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def install(verbose: bool = True) -> dict:
    """Steps 1 and 2 above; returns {"module.attr": "replaced" | reason}."""
    if DROPIN_DIR not in sys.path:
        sys.path.insert(0, DROPIN_DIR)
    report = {}
    swaps = (("local.units", "HighGammaExtractor", "dss_amd.units"),
             ("local.units", "DelayedLPCNetVocoder", "dss_amd.units"),
             ("local.units", "RecurrentNeuralDecodingModel", "dss_amd.units"),
             ("local.training", "AsynchronousSynthesisQueue", "dss_amd.synthesis_queue"))
    for mod_name, attr, ours in swaps:
        key = f"{mod_name}.{attr}"
        try:
            mod = importlib.import_module(mod_name)
        except Exception as e:      # the user's module needs packages this environment lacks
            report[key] = f"left alone: cannot import {mod_name} ({type(e).__name__}: {e})"
            continue
        if not hasattr(mod, attr):
            report[key] = f"left alone: {mod_name} has no {attr}"
            continue
        if attr == "RecurrentNeuralDecodingModel":       # wrapped, not replaced: a subclass of the user's own class
            setattr(mod, attr, importlib.import_module(ours).gpu_decoding_unit(getattr(mod, attr)))
        else:
            setattr(mod, attr, getattr(importlib.import_module(ours), attr))
        report[key] = "replaced"
    if verbose:
        for k, v in report.items():
            print(f"dss_amd.run: {k}: {v}", file=sys.stderr)
    return report


def main(argv=None) -> None:
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit("usage: python -m dss_amd.run <reference script.py> [its arguments]")
    script = os.path.abspath(argv[0])
    sys.path.insert(0, os.path.dirname(script))          # what `python script.py` would have done
    install()
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
