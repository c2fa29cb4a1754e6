"""CPU oracles (test infrastructure).  See the header of each module."""
