"""Convolution solver selection for the MIOpen-backed layers of the path (stem, decoder, MSMM conv branches).

The reference turns on ``cudnn.benchmark`` (run_training.py:123-125).  On ROCm that flag is MIOpen's exhaustive find:
every candidate solver of every convolution is compiled and timed at first use -- about 20 minutes for the 114
convolution problems of the 256x256 train step on a fresh box with an empty user database.  The result of that search
(a find-db and a perf-db, ~100 KB of text, produced once with MLAGG_MIOPEN_FIND=1 as tools/miopen_find.sh shows) is
kept next to this file; ``use_tuned_convolutions`` points MIOpen at a private copy of it and selects FAST find mode:
problems in the database get their measured-best solver, anything else falls back to MIOpen's immediate-mode
heuristic instead of a search.  Worth 2.6 % of the step at config 2 (55.9 -> 54.5 ms).
"""
import atexit
import glob
import os
import shutil
import tempfile

import torch

DB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")
TUNED_PATCH = (256, 256)          # the committed records: fp32 convolutions of the 256 x 256, batch-10 train step
_PRIVATE = []


def _cleanup():
    for d in _PRIVATE:
        shutil.rmtree(d, ignore_errors=True)


def use_tuned_convolutions(enabled=True):
    """Call before the first convolution of the process.  Returns the database directory in use (None: exhaustive
    find requested through MLAGG_MIOPEN_FIND=1, or tuning disabled through MLAGG_MIOPEN_TUNED=0 / ``enabled=False``).
    The committed database covers the fp32 convolutions of the 256x256, batch-10 train step; for other shapes pass
    ``enabled=False`` (immediate-mode solver choice, no find of any kind: a find-db MISS in FAST mode was seen to run
    MIOpen's naive reference convolutions for minutes on 224x224 fp32 problems)."""
    if not enabled:
        torch.backends.cudnn.benchmark = False
        return None
    if os.environ.get("MLAGG_MIOPEN_FIND", "0") == "1":
        torch.backends.cudnn.benchmark = True              # full search; MIOPEN_USER_DB_PATH is the caller's business
        return None
    files = glob.glob(os.path.join(DB_DIR, "*.txt"))
    if os.environ.get("MLAGG_MIOPEN_TUNED", "1") != "1" or not files:
        torch.backends.cudnn.benchmark = False
        return None
    private = tempfile.mkdtemp(prefix="mlagg_miopen_db_")  # per process: MIOpen rewrites the files it opens
    if not _PRIVATE:
        atexit.register(_cleanup)
    _PRIVATE.append(private)
    for f in files:
        if f.endswith(".ufdb.txt"):
            # MIOpen re-validates a find-db record against its KERNEL cache, which is empty on a fresh box: the first
            # listed solver whose binary is missing makes it re-time every solver of the record ("Find-db regenerating"),
            # among them the naive reference convolutions (up to 0.4 s per launch: 17 s of the first step).  They are
            # never the fastest, so they are dropped from the records and switched off for the re-timing.
            with open(f) as src, open(os.path.join(private, os.path.basename(f)), "w") as dst:
                for line in src:
                    key, _, rec = line.rstrip("\n").partition("=")
                    keep = [e for e in rec.split(";") if e and not e.startswith("ConvDirectNaiveConv")]
                    if keep:
                        dst.write(key + "=" + ";".join(keep) + "\n")
        else:
            shutil.copy(f, private)
    for d in ("FWD", "BWD", "WRW"):
        os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + d, "0")
    os.environ["MIOPEN_USER_DB_PATH"] = private
    os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")       # database hit -> tuned solver, miss -> heuristic, never a search
    torch.backends.cudnn.benchmark = True
    return private
