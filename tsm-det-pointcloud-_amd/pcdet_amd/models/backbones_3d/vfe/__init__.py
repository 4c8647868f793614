from .dynamic_mean_vfe import DynamicMeanVFE
from .mean_vfe import MeanVFE
from .vfe_template import VFETemplate

__all__ = {
    'VFETemplate': VFETemplate,
    'MeanVFE': MeanVFE,
    'DynamicMeanVFE': DynamicMeanVFE,
}
