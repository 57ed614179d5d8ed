// photon_version.hip - photon_version(): which build this is.  photon_amd/build.py compiles this unit with
// -DPHOTON_BUILD_ID="<git commit>[-dirty] <flags>" where <flags> is "default" or the non-default -DPHOTON_* switches the
// library was built with (tools/ab.sh variants): a variant library can be told from the shipped one once it is loaded, and
// bench.py prints the string in its line.
#ifndef PHOTON_BUILD_ID
#define PHOTON_BUILD_ID "unversioned build (compiled without photon_amd/build.py)"
#endif

extern "C" const char *photon_version(void) { return "photon-amd 0.5 (gfx950, HIP) " PHOTON_BUILD_ID; }
