"""Batched iLQG backward/forward pass for MI355X behind the DDP-Generator C API.

The directory name contains a hyphen, so load it with `load_package()` from
`__graft_entry__.py` (which registers it as `ddp_generator_amd`)."""

from . import synth  # noqa: F401,E402
from . import dist  # noqa: F401,E402
from . import ilqg  # noqa: F401,E402
from . import evidence  # noqa: F401,E402
