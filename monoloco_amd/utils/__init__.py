"""Host mirror of ``monoloco.utils`` for the hot path (reference monoloco/utils/__init__.py:2-13):
only what the keypoint->3D path and ``Loco.post_process`` touch."""
from .camera import back_correct_angles, get_keypoints, pixel_to_camera, to_cartesian, xyz_from_distance
from .iou import calculate_iou, get_iou_matches, get_iou_matches_matrix, get_iou_matrix, open_annotations, \
    reorder_matches

__all__ = ['pixel_to_camera', 'get_keypoints', 'xyz_from_distance', 'to_cartesian', 'back_correct_angles',
           'calculate_iou', 'get_iou_matrix', 'get_iou_matches', 'get_iou_matches_matrix', 'reorder_matches',
           'open_annotations']
