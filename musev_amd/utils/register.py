"""Name -> class table used to pick block / attention-processor implementations by the strings of a model config.

Interface of the reference's registry (musev/utils/register.py: ``@Model_Register.register`` with or without an alias,
lookup with ``[]``, ``in``, ``keys()``) so that modules written against it keep working; the implementation is this
package's own: a thin mapping with one explicit ``add`` entry point that the decorator forms delegate to."""
from __future__ import annotations

import logging
from typing import Callable, Dict, Iterable, Optional, Union

_log = logging.getLogger(__name__)


class Register:
    def __init__(self, registry_name: str):
        self.registry_name = registry_name
        self._entries: Dict[str, Callable] = {}

    # -- registration ---------------------------------------------------------------------------------------------------
    def add(self, obj: Callable, alias: Optional[str] = None) -> Callable:
        """file ``obj`` under, in order of preference: its own ``name`` attribute, ``alias``, its ``__name__``"""
        if not callable(obj):
            raise TypeError(f"registry {self.registry_name!r} holds callables (classes / functions), got {obj!r}")
        label = vars(obj).get("name") or alias or obj.__name__
        if label in self._entries:
            _log.warning("registry %s: %s registered again, the new entry wins", self.registry_name, label)
        self._entries[label] = obj
        return obj

    def register(self, target: Union[str, Callable]):
        """``@reg.register`` or ``@reg.register("alias")``"""
        if isinstance(target, str):
            return lambda obj: self.add(obj, alias=target)
        return self.add(target)

    def __setitem__(self, alias: Optional[str], obj: Callable) -> None:
        self.add(obj, alias=alias)

    # -- lookup ---------------------------------------------------------------------------------------------------------
    def __getitem__(self, label: str) -> Callable:
        try:
            return self._entries[label]
        except KeyError:
            raise KeyError(f"{label!r} is not registered in {self.registry_name!r} (known: {sorted(self._entries)})") from None

    def __contains__(self, label: str) -> bool:
        return label in self._entries

    def keys(self) -> Iterable[str]:
        return self._entries.keys()
