"""MD inference driver: mirror of kgcnn.moldyn.base.MolDynamicsModelPredictor."""
from .base import MolDynamicsModelPredictor  # noqa: F401
