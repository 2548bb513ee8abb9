"""Evaluation helpers with the module names of the reference's helpers/ directory."""
