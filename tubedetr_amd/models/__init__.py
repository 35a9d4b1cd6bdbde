"""``from tubedetr_amd.models import build_model`` replaces ``from models import build_model`` (models/__init__.py:1-4)."""
from .tubedetr import build


def build_model(args):
    return build(args)
