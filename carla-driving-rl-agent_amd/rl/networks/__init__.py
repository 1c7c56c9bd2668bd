"""`Network` base class: the surface PPOAgent drives (reference rl/networks/networks.py:13-46)."""


class Network:
    def __init__(self, agent):
        self.agent = agent

    def predict(self, *args, **kwargs):
        raise NotImplementedError

    def act(self, *args, **kwargs):
        raise NotImplementedError

    def reset(self):
        pass

    def trainable_variables(self):
        raise NotImplementedError

    def set_weights(self, weights):
        raise NotImplementedError

    def get_weights(self):
        raise NotImplementedError

    def load_weights(self):
        raise NotImplementedError

    def save_weights(self):
        raise NotImplementedError

    def summary(self):
        pass
