"""World registry (reference: numbotics/physics/world.py:24-64,213-259).

Only the object registries the hot path reads (``_static_objects`` / ``_dynamic_objects``, iterated at
robots/arm.py:228,564) are kept.  There is no PyBullet client, stepping, visualiser or
constraint store: simulation is out of scope.  ``World.pool`` (world.py:98-156)
exists upstream to clone PyBullet worlds for CPU threads; the device path batches instead, so it is
not reproduced.
"""
import itertools
import weakref
from typing import Any

WORLD_INSTANCES: dict = {}
SELECTED_WORLD = None


def get_world(name=None):
    global SELECTED_WORLD
    if name is None:
        name = SELECTED_WORLD
    if name is None and len(WORLD_INSTANCES) == 0:
        name = 'world_0'
    if name is None:
        raise ValueError('There should always be a selected World instance if a World exists')
    if name not in WORLD_INSTANCES:
        World(name=name)
    SELECTED_WORLD = name
    return WORLD_INSTANCES[name]


class World:

    def __init__(self, name=None, visualize: bool = False):
        global SELECTED_WORLD
        if visualize:
            raise NotImplementedError("visualisation is out of scope for the MI355X hot path")
        if name in WORLD_INSTANCES:
            raise ValueError(f'World with name {name} already exists')
        self._name = f'world_{len(WORLD_INSTANCES)}' if name is None else name
        WORLD_INSTANCES[self._name] = self
        SELECTED_WORLD = self._name
        # weak, like upstream: a body that goes out of scope leaves the world.
        self._static_objects = weakref.WeakValueDictionary()
        self._dynamic_objects = weakref.WeakValueDictionary()
        self._entity_map = weakref.WeakValueDictionary()
        self._ids = itertools.count()
        self._vis = None
        self._revision = 0          # bumped whenever the scene changes; device scenes key on it
        self.gravity = None

    @property
    def name(self):
        return self._name

    def _next_id(self) -> int:
        return next(self._ids)

    def _touch(self):
        self._revision += 1

    def register(self, body):
        from .object import PhysicsObject
        from .chain import Chain
        if not isinstance(body, (PhysicsObject, Chain)):
            raise ValueError(f"Unknown body type: {type(body)}")
        if body._static:
            self._static_objects[body.name] = body
        else:
            self._dynamic_objects[body.name] = body
        self._entity_map[body._pyb_id] = body
        self._touch()

    def unregister(self, body):
        for reg in (self._static_objects, self._dynamic_objects):
            if body.name in reg:
                del reg[body.name]
        self._entity_map.pop(body._pyb_id, None)
        self._touch()

    def objects(self):
        return list(self._static_objects.values()) + list(self._dynamic_objects.values())

    def get_object(self, name: str, default: Any = None):
        from .chain import Chain
        if name in self._static_objects:
            return self._static_objects[name]
        if name in self._dynamic_objects:
            return self._dynamic_objects[name]
        for obj in list(self._dynamic_objects.values()) + list(self._static_objects.values()):
            if isinstance(obj, Chain):
                for link in obj._links:
                    if link.name == name:
                        return link
        return default

    def step(self):
        """No dynamics here; kept so reference scripts that call ``world.step()`` still run."""
        return None


def _reset_worlds():
    """Test helper: forget every world."""
    global SELECTED_WORLD
    WORLD_INSTANCES.clear()
    SELECTED_WORLD = None
