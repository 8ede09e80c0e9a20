"""Library-GEMM algorithm choice for the dense products that stay on rocBLAS / hipBLASLt (the small-M Linear layers of
encoder stages 2 / 3, attention side projections: everything `torch.mm` / `addmm` serves on the path).

PyTorch's TunableOp times the rocBLAS default and the hipBLASLt algorithms of a GEMM shape at first use.  The result of one
such run on the config-2 train step (42 shapes, 4 KB of text: `gemm_db/`) is committed; ``use_tuned_gemms`` loads it with
tuning switched OFF, so a shape in the table gets its measured-best algorithm and any other shape the library default -- no
search at run time.  Worth 0.8 % of the step (228.9 -> 230.8 images/s, same box).  The table carries the library versions it
was made with; PyTorch ignores it when they differ.  Regenerate with
    PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=out.csv python bench.py
"""
import atexit
import glob
import os
import shutil
import tempfile

import torch

DB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_db")


def use_tuned_gemms(enabled=True):
    """Call before the first GEMM of the process.  Returns the table in use (None: disabled through ``enabled=False`` /
    MLAGG_GEMM_TUNED=0, no table, or the user drives TunableOp through its own environment variables)."""
    files = sorted(glob.glob(os.path.join(DB_DIR, "*.csv")))
    if not enabled or not files or os.environ.get("MLAGG_GEMM_TUNED", "1") != "1" or "PYTORCH_TUNABLEOP_ENABLED" in os.environ:
        return None
    if not torch.cuda.is_available():
        return None
    tmp = tempfile.mkdtemp(prefix="mlagg_gemm_db_")
    atexit.register(shutil.rmtree, tmp, ignore_errors=True)
    private = os.path.join(tmp, os.path.basename(files[0]))
    shutil.copy(files[0], private)                    # never hand TunableOp the committed file
    tn = torch.cuda.tunable
    tn.enable(True)
    tn.tuning_enable(False)
    if hasattr(tn, "write_file_on_exit"):
        tn.write_file_on_exit(False)                  # nothing is tuned here, so there is nothing to write back (newer PyTorch only)
    tn.set_filename(private)
    ok = tn.read_file(private)                        # False when the Validator rows (library versions) do not match
    if not ok or not tn.get_results():
        tn.enable(False)
        return None
    from . import ops
    ops.GEMM_TABLE_LOADED[0] = True                   # dispatch thresholds against the library were measured with this table (ops.wgrad_min_rows)
    return private
