from .agents import Agent
from .ppo import PPOAgent, PPOMemory
