"""Append-only run log, same file and call surface as the reference's ``Logger`` (cell_type_annotation/logger.py:4-20):
``<main_dir>/results/log.txt``, ``log(message)``, ``log_all_hyperparameters(dict)``, ``close()``."""
import os
import time


class Logger:
    def __init__(self, main_dir):
        out_dir = os.path.join(main_dir, "results")
        os.makedirs(out_dir, exist_ok=True)
        self.log_file_path = os.path.join(main_dir, "results/log.txt")
        self.log_file = open(self.log_file_path, "w")
        self.log_file.write("Log file created at {}\n".format(time.ctime()))

    def log(self, message):
        self.log_file.write(str(message) + "\n")

    def log_all_hyperparameters(self, hyperparameters):
        self.log_file.write("Hyperparameters:\n")
        for name, value in hyperparameters.items():
            self.log_file.write(f"{name}: {value}\n")

    def close(self):
        self.log_file.close()
