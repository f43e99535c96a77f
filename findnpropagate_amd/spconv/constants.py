"""spconv.constants stand-in: pcdet/utils/spconv_utils.py:4-5 sets SPCONV_USE_DIRECT_TABLE."""
SPCONV_USE_DIRECT_TABLE = False
