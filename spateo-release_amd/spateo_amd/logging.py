"""Stand-in for ``spateo.logging.logger_manager`` (``spateo/logging.py:1-3`` -> ``spateo/external/lack.py:255-365``):
only the ``lm.main_*`` call semantics the morphofield wrappers use are kept (``sparsevfc.py:174,181,214,225,237``)."""
from __future__ import annotations

import logging as _logging

_logger = _logging.getLogger("spateo_amd")


class _LoggerManager:
    @staticmethod
    def _fmt(msg, indent_level=1):
        return "|" + "-" * (4 * max(int(indent_level) - 1, 0) + 4) + "> " + str(msg)

    def main_info(self, msg, indent_level=1):
        _logger.info(self._fmt(msg, indent_level))

    def main_warning(self, msg, indent_level=1):
        _logger.warning(self._fmt(msg, indent_level))

    def main_debug(self, msg, indent_level=1):
        _logger.debug(self._fmt(msg, indent_level))

    def main_finish_progress(self, progress_name=""):
        _logger.info(self._fmt(f"[{progress_name}] finished", 1))


logger_manager = _LoggerManager()
