"""`import pcdet` compatibility alias (SURVEY.md §8b boundary B1): the MI355X-native `pcdet_amd` package under the name the
reference's entry points import — `from pcdet.models import build_network, model_fn_decorator`
(tools/train.py:14-17, tools/train_utils/train_utils.py:8), `from pcdet.utils import common_utils, commu_utils`.
Put this directory (tsm-det-pointcloud-_amd/compat) on PYTHONPATH together with its parent.

`pcdet.X` resolves to THE SAME module object as `pcdet_amd.X` (a meta-path finder, not a second copy of the package), so
registries, isinstance checks and monkey-patches agree whichever name a caller used."""
import importlib
import importlib.abc
import importlib.util
import sys

_REAL = "pcdet_amd"


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname == __name__ or fullname.startswith(__name__ + "."):
            return importlib.util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len(__name__):])

    def exec_module(self, module):      # already executed under its real name
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
sys.modules[__name__] = importlib.import_module(_REAL)
