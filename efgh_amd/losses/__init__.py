from .efghloss import EFGHCriterion  # noqa: F401  (looked up by name, reference main.py:129)
