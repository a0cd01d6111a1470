"""manga_ocr - MI355X-native drop-in for the recogniser the reference imports at
``src/core/config.py:431-436`` (``from manga_ocr import MangaOcr``).

Put ``<repo>/manga-ocr_amd`` on ``sys.path`` (or ``pip install -e`` it) and the reference's
``src/core`` needs no edits.  See INTEGRATION.md."""
from .ocr import MangaOcr  # noqa: F401
from .text import post_process  # noqa: F401

__all__ = ["MangaOcr", "post_process"]
__version__ = "0.1.0"
