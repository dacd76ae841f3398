from .scheduling_ddim import DDIMScheduler  # noqa: F401
from .scheduling_euler_discrete import EulerDiscreteScheduler  # noqa: F401
