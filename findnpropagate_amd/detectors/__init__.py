from .detector3d_template import Detector3DTemplate, all_class_names, known3_labels, known6_labels

__all__ = ["Detector3DTemplate", "all_class_names", "known3_labels", "known6_labels"]
