"""The 3-D `sdf_*` evaluators by their reference names (reference cores/sdf_3D.py). Each is a
callable `sdf_*(co, *params) -> (N,) field` that runs on the GPU, and is accepted wherever the
reference takes an SDF function (`GenericGeometry(sdf_sphere, 0.5)`)."""
from .. import _prims as _P

_NAMES = ["sdf_x", "sdf_y", "sdf_z", "sdf_sphere", "sdf_cylinder", "sdf_box", "sdf_torus", "sdf_chainlink",
          "sdf_braid", "sdf_arc_3d", "sdf_plane", "sudf_plane", "sdf_segment_3d", "sdf_cone",
          "sdf_oriented_infinite_cone", "sdf_infinite_cone", "sdf_solid_angle", "sdf_triangle_3d", "sdf_quad_3d",
          "sdf_segmented_curve_3d", "sdf_segmented_line_3d", "sdf_parametric_curve_3d", "sdf_point_cloud_3d"]
globals().update({n: _P.get(n) for n in _NAMES})
__all__ = list(_NAMES)
