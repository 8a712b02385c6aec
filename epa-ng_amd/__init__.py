# The real package body; imported through the `epa_ng_amd` alias (see ../epa_ng_amd/__init__.py).
