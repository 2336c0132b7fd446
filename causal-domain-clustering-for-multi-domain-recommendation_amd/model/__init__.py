"""Host-side mirror of the reference's model/ registry (run.py:15-26 imports these names)."""
