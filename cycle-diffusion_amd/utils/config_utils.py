"""INI experiment configs -> nested argument objects.

Same contract as the reference's utils/config_utils.py:8-78 so that config/experiments/*.cfg files
drive this engine unchanged: sections become attributes, values are parsed int -> float -> bool ->
None -> JSON -> str, unknown attributes read as None, and iterating a section yields its
(key, value) pairs in sorted order (get_gan_wrapper turns the [gan] section into kwargs that way).
"""
import configparser
import json
import os


class Args:
    def __init__(self):
        object.__setattr__(self, "_items", {})

    def __getattr__(self, name):  # only called for names not found normally
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return self._items.get(name)

    def __setattr__(self, name, value):
        if value is not None:
            self._items[name] = value

    def __delattr__(self, name):
        self._items.pop(name, None)

    def __iter__(self):
        return iter(sorted(self._items.items()))

    def __len__(self):
        return len(self._items)

    def __contains__(self, name):
        return name in self._items

    def __repr__(self):
        return "Args(%r)" % (dict(self._items),)


def parse_string(string):
    for cast in (int, float):
        try:
            return cast(string)
        except ValueError:
            pass
    if string in ("True", "true"):
        return True
    if string in ("False", "false"):
        return False
    if string in ("none", "None"):
        return None
    try:
        return json.loads(string)
    except json.decoder.JSONDecodeError:
        pass
    return string.strip("\"'")


def get_config(cfg_name, config_root="config"):
    """cfg_name is relative to `config_root` (the reference resolves against ./config, :68)."""
    path = cfg_name if os.path.isabs(cfg_name) else os.path.join(config_root, cfg_name)
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    parser = configparser.ConfigParser()
    parser.read(path)
    args = Args()
    for section in parser.sections():
        sec = Args()
        for key, value in parser.items(section):
            setattr(sec, key, parse_string(value))
        setattr(args, section, sec)
    return args
