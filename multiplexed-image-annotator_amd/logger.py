"""Run log with the file name, wording and call surface of the reference's ``Logger`` (cell_type_annotation/logger.py:4-20):
``<main_dir>/results/log.txt``; ``log(message)``, ``log_all_hyperparameters(dict)``, ``close()``; attributes ``log_file_path`` and
``log_file`` as other code may read them.  Lines are flushed as they are written, so a crashed run still leaves its log."""
import os
import time

_LOG_NAME = "log.txt"


class Logger:
    def __init__(self, main_dir):
        folder = os.path.join(main_dir, "results")
        if not os.path.isdir(folder):
            os.makedirs(folder, exist_ok=True)
        self.log_file_path = os.path.join(folder, _LOG_NAME)
        self.log_file = open(self.log_file_path, "w", buffering=1)
        self._emit("Log file created at " + time.ctime())

    def _emit(self, line) -> None:
        print(line, file=self.log_file)

    def log(self, message) -> None:
        self._emit(message if isinstance(message, str) else str(message))

    def log_all_hyperparameters(self, hyperparameters) -> None:
        self._emit("Hyperparameters:")
        for name in hyperparameters:
            self._emit("%s: %s" % (name, hyperparameters[name]))

    def close(self) -> None:
        if not self.log_file.closed:
            self.log_file.close()
