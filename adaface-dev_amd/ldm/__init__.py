"""Mirror of the reference package path of the same name (only the hot-path modules)."""
