"""Caller of the mapping hot path: mirror of ``MappingModule.step`` (/root/reference/nerf_vo/mapping/
mapping_module.py:35-55) without the multiprocessing plumbing around it.  It decides, for every tick of the
process loop, whether the mapper trains: every queue item is handed to the mapper; between items the mapper
free-runs for at most ``mapping_iterations / num_keyframes`` idle ticks (so that training keeps pace with the
incoming keyframes), and without limit once the last frame has arrived."""
from __future__ import annotations


class MappingModule:
    def __init__(self, method, mapping_iterations: int, num_keyframes: int):
        self.method = method                      # Nerfstudio / InstantNGP mirror: callable(input=...), .is_shut_down
        self.mapping_iterations = mapping_iterations
        self.num_keyframes = num_keyframes
        self.is_receving_data = True              # (sic) attribute name of the reference
        self.last_received_data = 0
        self.step_counter = 0
        self.shutdown = False

    def step(self, input: dict | None) -> tuple:
        self.step_counter += 1
        skip_step = False
        if input is None:
            if self.last_received_data < self.mapping_iterations / self.num_keyframes or not self.is_receving_data:
                self.method(input=input)
            else:
                skip_step = True
            self.last_received_data += 1
        else:
            self.method(input=input)
            self.last_received_data = 0
        if input is not None and input["last_frame"]:
            self.is_receving_data = False
        if self.method.is_shut_down:
            self.shutdown = True
        return None, skip_step
