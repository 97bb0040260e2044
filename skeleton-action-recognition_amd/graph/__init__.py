"""Drop-in for the reference's `graph` package (graph/__init__.py): `from graph.ntu_rgb_d import Graph`."""
from . import tools, ntu_rgb_d  # noqa: F401
