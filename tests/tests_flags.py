"""Flag sets used by the tests.  FLYLIGHT here is the kernels-only set (connected components,
no thinning) that the stage-by-stage parity tests were written against; the shipped set
(mutex watershed + thinning) is FLYLIGHT_SHIPPED."""
from patchperpix_amd.flags import FLYLIGHT_NOTHIN_CC as FLYLIGHT  # noqa: F401
from patchperpix_amd.flags import FLYLIGHT as FLYLIGHT_SHIPPED  # noqa: F401
from patchperpix_amd.flags import FLYLIGHT_CC  # noqa: F401
