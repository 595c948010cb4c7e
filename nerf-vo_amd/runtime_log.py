"""Per-step runtime log either side of the mapping step: the reference wraps every
module step in a ``PerformanceTracker`` (/root/reference/nerf_vo/multiprocessing/performance_tracker.py:5-24),
ships ``(process_name, 'runtime', step, seconds)`` tuples to its logging process and writes one
``runtime_<process>.csv`` (columns ``step,runtime``) per process at shut-down
(/root/reference/nerf_vo/multiprocessing/logging_module.py:21-58).

Same tuples, same CSV.  One conscious difference, off by default: the reference times host enqueue only (it never
synchronises the GPU); ``synchronize=True`` drains the device before both time stamps so the
mapping row is the duration of the work rather than of its launch."""
from __future__ import annotations

import csv
import os
from time import time


class PerformanceTracker:
    def __init__(self, process_name: str, logging_queue, step: int, synchronize: bool = False, device=None) -> None:
        self.process_name = process_name
        self.logging_queue = logging_queue
        self.step = step
        self.start_time = 0.0
        self.runtime = 0.0
        self._sync = None
        if synchronize:
            import torch

            self._sync = lambda: torch.cuda.synchronize(device)

    def __enter__(self):
        if self._sync is not None:
            self._sync()
        self.start_time = time()
        return self

    def __exit__(self, *args) -> None:
        if self._sync is not None:
            self._sync()
        self.runtime = time() - self.start_time

    def submit(self) -> None:
        if self.logging_queue is not None:
            self.logging_queue.put((self.process_name, "runtime", self.step, self.runtime))


class RuntimeLog:
    """The logging process's bookkeeping: ``step(events)`` consumes queue tuples, ``shut_down(dir_result)``
    writes the CSVs.  A 'shutdown' field raises the flag the reference's module loop polls."""

    def __init__(self) -> None:
        self.logs: dict[str, list] = {}
        self.shutdown = False

    def drain(self, queue) -> list:
        events = []
        while True:
            try:
                events.append(queue.get(block=False))
            except Exception:
                break
        return events

    def step(self, events) -> None:
        for process_name, field, step, runtime in events:
            if field == "shutdown":
                self.shutdown = True
            elif field == "runtime":
                self.logs.setdefault(process_name, []).append({"step": step, "runtime": runtime})

    def shut_down(self, dir_result: str) -> list:
        os.makedirs(dir_result, exist_ok=True)
        written = []
        for key, rows in self.logs.items():
            path = f"{dir_result}/runtime_{key}.csv"
            with open(path, "w", newline="") as file:
                writer = csv.DictWriter(file, fieldnames=["step", "runtime"], lineterminator="\n")
                writer.writeheader()
                writer.writerows(rows)
            written.append(path)
        return written
