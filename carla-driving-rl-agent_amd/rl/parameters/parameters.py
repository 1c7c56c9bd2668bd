"""Step-dependent scalars (learning rates, clip ratio, entropy coefficient, advantage scale).

Same surface as the reference's rl/parameters/parameters.py (DynamicParameter.create / __call__ /
serialize / load / on_episode, ConstantParameter, ScheduleWrapper, ExponentialDecay, StepDecay,
PolynomialDecay), with the Keras LearningRateSchedule objects replaced by plain callables
step -> value.  Host-side only: the values travel to the device in the cdrl_hparams block."""
import math


class DynamicParameter:
    def __init__(self):
        self.value = 0
        self.step = 0

    @staticmethod
    def create(value, **kwargs):
        if isinstance(value, (float, int)) and not isinstance(value, bool):
            return ConstantParameter(float(value))
        if isinstance(value, DynamicParameter):
            return value
        if callable(value):
            return ScheduleWrapper(schedule=value, **kwargs)
        raise TypeError(f'cannot make a DynamicParameter out of {type(value).__name__}')

    def __call__(self, *args, **kwargs):
        return self.value

    def serialize(self) -> dict:
        return dict(step=int(self.step))

    def on_episode(self):
        self.step += 1

    def load(self, config: dict):
        self.step = config.get('step', 0)

    def get_config(self) -> dict:
        return {}


class ConstantParameter(DynamicParameter):
    def __init__(self, value: float):
        super().__init__()
        self.value = value

    def serialize(self) -> dict:
        return {}


class ScheduleWrapper(DynamicParameter):
    """value(step) = max(min_value, schedule(step)); the step advances once per episode."""

    def __init__(self, schedule, min_value=1e-4):
        super().__init__()
        self.schedule = schedule
        self.min_value = min_value
        self.value = max(min_value, float(schedule(0)))

    def __call__(self, *args, **kwargs):
        self.value = max(self.min_value, float(self.schedule(self.step)))
        return self.value

    def get_config(self) -> dict:
        return dict(getattr(self.schedule, 'config', {}))


def _exponential(initial, decay_steps, decay_rate, staircase):
    def fn(step):
        p = step / decay_steps
        if staircase:
            p = math.floor(p)
        return initial * decay_rate ** p
    fn.config = dict(initial_learning_rate=initial, decay_steps=decay_steps, decay_rate=decay_rate, staircase=staircase)
    return fn


def _polynomial(initial, end, decay_steps, power, cycle):
    def fn(step):
        ds = decay_steps
        if cycle:
            ds = decay_steps * max(1.0, math.ceil(step / decay_steps))
        else:
            step = min(step, decay_steps)
        return (initial - end) * (1.0 - step / ds) ** power + end
    fn.config = dict(initial_learning_rate=initial, end_learning_rate=end, decay_steps=decay_steps, power=power, cycle=cycle)
    return fn


class ExponentialDecay(ScheduleWrapper):
    def __init__(self, initial_value: float, decay_steps: int, decay_rate: float, staircase=False, min_value=0.0):
        super().__init__(_exponential(initial_value, decay_steps, decay_rate, staircase), min_value=min_value)


class StepDecay(ScheduleWrapper):
    def __init__(self, initial_value: float, decay_steps: int, decay_rate: float, min_value=1e-4):
        super().__init__(_exponential(initial_value, decay_steps, decay_rate, True), min_value=min_value)


class PolynomialDecay(ScheduleWrapper):
    def __init__(self, initial_value: float, end_value: float, decay_steps: int, power=1.0, cycle=False):
        super().__init__(_polynomial(initial_value, end_value, decay_steps, power, cycle))
