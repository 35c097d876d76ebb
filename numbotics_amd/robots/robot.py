"""Base class (reference: numbotics/robots/robot.py)."""


class Robot:
    def __init__(self, chain):
        self._chain = chain
