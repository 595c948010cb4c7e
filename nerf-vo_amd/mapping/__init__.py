"""Host-side mirror of the reference's mapping interface (nerf_vo/mapping + the nerfstudio objects it
touches, SURVEY.md section 8b "outer boundary"), implemented over the native engine."""
from .cameras import Cameras, CameraType, RayBundle  # noqa: F401
from .dataset import DynamicDataManager, DynamicDataManagerConfig, DynamicDataset  # noqa: F401
