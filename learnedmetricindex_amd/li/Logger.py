"""Logging mixin with the reference's logger naming and format (reference: li/Logger.py:1-18)."""
import logging

_FORMAT = "[%(asctime)s][%(levelname)-5.5s][%(name)-.20s] %(message)s"


def get_logger_config() -> str:
    return _FORMAT


def remove_logger_handlers() -> None:
    root = logging.getLogger()
    for handler in list(root.handlers):
        root.removeHandler(handler)


class Logger:
    """Gives a class a `self.logger` named `<module>.<ClassName>` (Logger.py:13-18)."""

    @property
    def logger(self) -> logging.Logger:
        cls = type(self)
        logging.basicConfig(level=logging.INFO, format=_FORMAT)
        return logging.getLogger(f"{cls.__module__}.{cls.__name__}")
