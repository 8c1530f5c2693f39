from .chamfer_distance import ChamferDistance  # noqa: F401
