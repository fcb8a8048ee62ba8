// wx_buildinfo.hip -- wx_build_info(): what this binary was built from (mkbuildinfo.sh writes the string at build time).
#include "../../include/waveletsext_hip.h"
#include "wx_buildinfo_gen.h"

extern "C" const char *wx_build_info(void) { return WX_BUILD_INFO; }
