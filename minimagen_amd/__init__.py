"""minimagen_amd -- MI355X-native implementation of MinImagen's cascaded-diffusion sampling hot
path behind the reference's own ``Imagen`` / ``Unet`` API."""
__version__ = "0.1.0"

import os as _os

# The pipelined sampler keeps (call lanes x cascade stages) + the caller's stream busy -- five HIP streams for the two-stage cascade -- and
# HIP multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues.  Streams that share a queue with the stream feeding them pay
# ~16 us per graph node (profiles/r04_second_instance_slowdown.txt); eight queues leave none shared.  Only a default: an explicit setting
# wins, and it has no effect once the HIP runtime is initialised.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
