"""Enums on the hot path's API surface (mirror of reference splitp/enums.py:3-27).

Members take their own names as values, like the reference's `_generate_next_value_`."""
from enum import Enum


class FlatFormat(Enum):
    sparse = "sparse"
    reduced = "reduced"


class Method(Enum):
    flattening = "flattening"
    subflattening = "subflattening"
    distance = "distance"
    mutual_information = "mutual_information"
