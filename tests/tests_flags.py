from patchperpix_amd.flags import FLYLIGHT  # noqa: F401
