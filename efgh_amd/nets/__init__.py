from .efghbackbone import EFGHBackbone  # noqa: F401  (looked up by name, reference main.py:126)
