"""speaker-recognition_amd: MI355X-native GMM-UBM / i-vector sufficient-statistics path with the
method surface of sidekit.{Mixture, StatServer, FactorAnalyser} (reference: Anwarvic/Speaker-Recognition,
a SIDEKIT 1.3.2 fork).  Import through the repo-root shim:  `import sidekit_amd as sidekit`."""
from .engine import HipEngine, SkmError, default_engine, set_default_engine, load_library, LIB_PATH  # noqa: F401
from .idmap import IdMap, Ndx, Scores, Key  # noqa: F401
from .features_server import FeaturesServer  # noqa: F401
from .mixture import Mixture, sum_log_probabilities  # noqa: F401
from .statserver import StatServer  # noqa: F401
from .factor_analyser import FactorAnalyser  # noqa: F401
from .gmm_scoring import gmm_scoring, gmm_scoring_singleThread  # noqa: F401
from . import iv_scoring  # noqa: F401
from .iv_scoring import (cosine_scoring, mahalanobis_scoring, two_covariance_scoring, fast_PLDA_scoring,  # noqa: F401
                         PLDA_scoring, full_PLDA_scoring)
from . import dist  # noqa: F401
from . import h5lite  # noqa: F401

PARAM_TYPE = "float32"
STAT_TYPE = "float64"
__version__ = "0.1.0"
