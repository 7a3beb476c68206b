"""db_text_minimal_amd — MI355X-native DBNet hot path (model forward/backward,
DBLoss, per-step update) behind the call surface of huyhoang17/DB_text_minimal."""
from .losses import DBLoss  # noqa: F401
from .models import DBTextModel  # noqa: F401
from .optim import FusedAdam  # noqa: F401
from .train import DBTrainer  # noqa: F401
