from .carla_agent import CARLAgent, CARLAMemory, FakeCARLAEnvironment
from .networks import CARLANetwork, dynamics_layers
