"""String-keyed class registry with the reference's interface (musev/utils/register.py:6-44): classes register
themselves with ``@Model_Register.register`` and UNet3DConditionModel resolves block / processor classes by name."""
import logging

logger = logging.getLogger(__name__)


class Register:
    def __init__(self, registry_name):
        self._dict = {}
        self._name = registry_name

    def __setitem__(self, key, value):
        if not callable(value):
            raise Exception(f"Value of a Registry must be a callable!\nValue: {value}")
        if "name" in value.__dict__:
            key = value.name
        elif key is None:
            key = value.__name__
        if key in self._dict:
            logger.warning("Key %s already in registry %s." % (key, self._name))
        self._dict[key] = value

    def register(self, target):
        def add(key, value):
            self[key] = value
            return value

        if callable(target):
            return add(None, target)
        return lambda x: add(target, x)

    def __getitem__(self, key):
        return self._dict[key]

    def __contains__(self, key):
        return key in self._dict

    def keys(self):
        return self._dict.keys()
