"""Import alias: `carla-driving-rl-agent_amd/` (the product package directory, whose name is not a
valid Python identifier) is exposed as `carla_driving_rl_agent_amd`."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'carla-driving-rl-agent_amd')
__path__.insert(0, _real)
exec(compile(open(_os.path.join(_real, '__init__.py')).read(), _os.path.join(_real, '__init__.py'), 'exec'))
