from .simulate import forward, Simulator, group_measurements
from .transform import LinearTransform, Transform, rotation_matrix
