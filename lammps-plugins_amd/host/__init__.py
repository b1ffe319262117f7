"""Host-side plumbing around the hot path: system construction, decomposition, ctypes binding."""
