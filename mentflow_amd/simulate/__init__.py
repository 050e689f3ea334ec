from .simulate import forward, Simulator, group_measurements
from .transform import CompositeTransform, LinearTransform, MultipoleTransform, Transform, rotation_matrix
