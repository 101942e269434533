"""Host-side utilities the hot path touches: communicator, data log, trace points."""
