"""Experiment harnesses that define the benchmarked step (reference experiments/fig2a)."""
