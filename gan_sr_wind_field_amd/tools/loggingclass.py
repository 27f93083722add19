"""Process-global status log shared by every model object.

Mirrors the reference's ``tools/loggingclass.py:10-23``: the list is a *class*
attribute, so G, D and the GAN wrapper append to one buffer that the train
loop drains with ``get_new_status_logs()``.
"""
from typing import List


class GlobalLoggingClass:
    status_logs: List[str] = []

    def get_new_status_logs(self) -> List[str]:
        drained = list(self.status_logs)
        del self.status_logs[:]
        return drained
