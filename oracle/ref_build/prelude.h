// Build prelude for oracle/_ref (see oracle/ref_build/Makefile). NOT reference code.
// The reference's libSLR/defines.h has no working Linux branch for SLR_memalign
// (defines.h:105-108 expands to a bare identifier), so its own posix_memalign branch
// (defines.h:95-104, the OS X / OpenBSD one) is selected by overriding the platform
// macro for the duration of that one include.  System headers are pulled in first,
// with the true platform macros, so only defines.h sees the override.
#pragma once
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstdarg>
#include <cmath>
#include <cfloat>
#include <cstring>
#include <ctime>
#include <iostream>
#include <iomanip>
#include <string>
#include <sstream>
#include <array>
#include <vector>
#include <deque>
#include <map>
#include <set>
#include <stack>
#include <chrono>
#include <limits>
#include <algorithm>
#include <memory>
#include <functional>
#include <thread>
#include <unistd.h>
#pragma push_macro("__linux__")
#undef __linux__
#define __OpenBSD__ 1
#include "defines.h"
#undef __OpenBSD__
#pragma pop_macro("__linux__")
#ifdef SLR_ORACLE_RGB
// RGB mode is the reference's compile-time switch (defines.h:160, references.h:45-59).
#undef Use_Spectral_Representation
#endif
