// Library identity + tiny utilities.
#include "common.h"

extern "C" int dcc_version() { return 0; }  // returns 0 (DCC_OK) so that the generic int-rc check applies
