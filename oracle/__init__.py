"""Test-only CPU oracle. See oracle_np.py for the scope rules."""
