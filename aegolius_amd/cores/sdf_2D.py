"""The 2-D `sdf_*` evaluators by their reference names (reference cores/sdf_2D.py); see sdf_3D."""
from .. import _prims as _P

_NAMES = ["sdf_circle", "sdf_neu_circle", "sdf_box_2d", "sdf_segment_2d", "sdf_rounded_box_2d", "sdf_triangle_2d",
          "sdf_arc", "sdf_sector", "sdf_inf_sector", "sdf_ngon", "sdf_segmented_curve_2d", "sdf_segmented_line_2d",
          "sdf_polygon_2d", "sdf_parametric_curve_2d", "sdf_point_cloud_2d"]
globals().update({n: _P.get(n) for n in _NAMES})
__all__ = list(_NAMES)
