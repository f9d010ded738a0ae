"""Configuration helpers with the surface of the reference's `uav_ac/utils.py` (get_config :8-19,
parse_array :22-28): the INI file next to this module, `#` inline comments allowed."""
from __future__ import annotations

import ast
import configparser
import os

import numpy as np

CONFIG_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config.ini")


def get_config(path: str | None = None):
    """-> (DEFAULT section, SIM_FLIGHT section) of config.ini (or of `path`)."""
    parser = configparser.ConfigParser(inline_comment_prefixes="#")
    if not parser.read(path or CONFIG_FILE):
        raise FileNotFoundError(path or CONFIG_FILE)
    return parser["DEFAULT"], parser["SIM_FLIGHT"]


def parse_array(section, key: str) -> np.ndarray:
    """An entry holding a Python list literal, e.g. `limits = [[0, 0, 0], [10, 10, 10]]`, as an array."""
    return np.array(ast.literal_eval(section.get(key)))
