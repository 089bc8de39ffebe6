"""Scalar logging side service (reference: /root/reference/src/oprl/logging.py).
Out of the hot path; kept API-compatible so config scripts run unchanged.
tensorboard is optional (absent on the GPU box): the text ``<tag>.log`` files
are always written, the SummaryWriter only if importable.  Unlike the reference,
``log_scalars`` really logs (the reference's generator expression is never
consumed, logging.py:64-70)."""
from __future__ import annotations

import logging
import os
import shutil
import sys
from abc import ABC, abstractmethod
from datetime import datetime
from pathlib import Path
from typing import Callable, Protocol, runtime_checkable

try:  # pragma: no cover - optional dependency
    from torch.utils.tensorboard.writer import SummaryWriter  # type: ignore
except Exception:  # noqa: BLE001
    SummaryWriter = None


@runtime_checkable
class LoggerProtocol(Protocol):
    log_dir: Path

    def log_scalar(self, tag: str, value: float, step: int) -> None: ...

    def log_scalars(self, values: dict[str, float], step: int) -> None: ...


def get_logs_path(logdir: str, algo: str, env: str, seed: int) -> Path:
    stamp = datetime.now().strftime("%Y_%m_%d_%Hh%Mm%Ss")
    log_dir = Path(logdir) / algo / f"{algo}-env_{env}-seed_{seed}-{stamp}"
    logging.info(f"LOGDIR: {log_dir}")
    return log_dir


def create_stdout_logger(name: str | None = None) -> logging.Logger:
    if name is None:
        import inspect
        frame = inspect.currentframe().f_back
        name = os.path.splitext(os.path.basename(frame.f_code.co_filename))[0]
    lg = logging.getLogger(name)
    lg.setLevel(logging.INFO)
    return lg


logger = create_stdout_logger(__name__)


class BaseLogger(ABC):
    @abstractmethod
    def log_scalar(self, tag: str, value: float, step: int) -> None: ...

    def log_scalars(self, values: dict[str, float], step: int) -> None:
        for k, v in values.items():
            self.log_scalar(k, v, step)


class NullLogger(BaseLogger):
    """Drops everything (benchmarks, tests)."""

    def __init__(self, logdir: Path | str = ".") -> None:
        self.log_dir = Path(logdir)

    def log_scalar(self, tag: str, value: float, step: int) -> None:
        return None


class FileTxtLogger(BaseLogger):
    def __init__(self, logdir: Path | str) -> None:
        self.log_dir = Path(logdir)
        self.writer = SummaryWriter(str(logdir)) if SummaryWriter is not None else None

    def copy_source_code(self) -> None:
        dest = self.log_dir / "src"
        shutil.copytree(Path(__file__).parent, dest, dirs_exist_ok=True,
                        ignore=shutil.ignore_patterns("*.so", "__pycache__"))
        main_module = sys.modules.get("__main__")
        if main_module is not None and getattr(main_module, "__file__", None):
            shutil.copyfile(main_module.__file__, self.log_dir / Path(main_module.__file__).name)
        else:
            logger.warning("Failed to copy config file.")

    def log_scalar(self, tag: str, value: float, step: int) -> None:
        if self.writer is not None:
            self.writer.add_scalar(tag, value, step)
        path = self.log_dir / f"{tag}.log"
        path.parent.mkdir(parents=True, exist_ok=True)
        with open(path, "a") as f:
            f.write(f"{step} {value}\n")


def copy_exp_dir(log_dir: Path) -> None:
    """Snapshot of the package sources next to the logs (reference logging.py:42-46)."""
    dest = Path(log_dir) / "src"
    shutil.copytree(Path(__file__).parent, dest, dirs_exist_ok=True,
                    ignore=shutil.ignore_patterns("*.so", "__pycache__"))
    logger.info(f"Source copied into {dest}")


class TextLoggerFactory:
    """``factory(seed) -> FileTxtLogger`` under ``$OPRL_LOGS/<algo>/<env>/...``.  A class, not a closure:
    the multi-seed runner hands its factories to spawned processes, so they have to pickle."""

    def __init__(self, algo: str, env: str) -> None:
        self.algo, self.env = algo, env

    def __call__(self, seed: int) -> LoggerProtocol:
        root = os.environ.get("OPRL_LOGS", "logs")
        lg = FileTxtLogger(get_logs_path(logdir=root, algo=self.algo, env=self.env, seed=seed))
        lg.log_dir.mkdir(parents=True, exist_ok=True)
        lg.copy_source_code()
        return lg


def make_text_logger_func(algo: str, env: str) -> Callable[[int], LoggerProtocol]:
    """The reference's factory-of-a-factory (logging.py:49-56)."""
    return TextLoggerFactory(algo, env)
