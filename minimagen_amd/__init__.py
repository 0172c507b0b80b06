"""minimagen_amd -- MI355X-native implementation of MinImagen's cascaded-diffusion sampling hot
path behind the reference's own ``Imagen`` / ``Unet`` API."""
__version__ = "0.1.0"
