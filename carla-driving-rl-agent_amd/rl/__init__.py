from . import utils, spaces, parameters
from .parameters import DynamicParameter
from .agents import Agent, PPOAgent, PPOMemory
