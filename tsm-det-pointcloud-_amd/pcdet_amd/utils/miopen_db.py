"""Tuned MIOpen user database for the dense tail (see tsm-det-pointcloud-_amd/miopen_db/README.md).

The BEV backbone's convolutions are stock MIOpen kernels; which solver and which tile parameters MIOpen uses in immediate
mode (`torch.backends.cudnn.benchmark = False`, the reference's setting) comes from its databases.  `use_tuned_db()` makes the
entries MIOpen's own tuner found for this model's convolution problems visible to the process — call it before the first
convolution runs (bench.py does, first thing in main())."""
import glob
import os
import shutil
import tempfile

_DB_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "miopen_db")


def use_tuned_db(force=False):
    """Point MIOPEN_USER_DB_PATH at a scratch copy of the shipped database (MIOpen also writes there).  Leaves an
    existing MIOPEN_USER_DB_PATH alone unless force=True.  Returns the directory used, or None."""
    if os.environ.get("MIOPEN_USER_DB_PATH") and not force:
        return os.environ["MIOPEN_USER_DB_PATH"]
    files = glob.glob(os.path.join(_DB_DIR, "*.txt"))
    if not files:
        return None
    scratch = tempfile.mkdtemp(prefix="spx_miopen_db_")
    for f in files:
        shutil.copy(f, scratch)
    os.environ["MIOPEN_USER_DB_PATH"] = scratch
    return scratch
