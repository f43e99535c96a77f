"""Registry entry the reference looks up by NAME (pcdet/models/dense_heads/__init__.py:38-75)."""
from .clip_box_classification import CLIPBoxClassification
from .frustum_proposals_v1 import FrustumProposerOG

__all__ = {"FrustumProposerOG": FrustumProposerOG, "CLIPBoxClassification": CLIPBoxClassification}
