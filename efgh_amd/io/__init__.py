"""Rows "next" of SURVEY.md §8(f): on-disk formats and checkpoint interchange around the hot path."""
