"""Mirror of the reference package ``adaface`` (hot-path modules only)."""
