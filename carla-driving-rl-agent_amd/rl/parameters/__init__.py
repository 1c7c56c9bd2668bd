from .parameters import (DynamicParameter, ConstantParameter, ScheduleWrapper, ExponentialDecay, StepDecay,
                         PolynomialDecay)
