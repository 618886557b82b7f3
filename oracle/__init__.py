"""Test infrastructure: CPU restatement of the reference hot path (see istvt_ref.py)."""
