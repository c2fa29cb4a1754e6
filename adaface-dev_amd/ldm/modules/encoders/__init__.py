"""Text-conditioning encoders (reference ldm/modules/encoders)."""
