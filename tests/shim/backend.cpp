// The translation unit INTEGRATION.md §2 prescribes (src/hip_backend.cpp in the reference tree): the reference's declarations,
// then the shim.  It calls none of the functions itself -- they must be emitted with external linkage all the same.
#include "rm_contract.hpp" // in the reference tree: "core.h", "imgproc.h", "objdetect.h", "mobility.h"
#include "rmcv_shim.hpp"
#include <type_traits>
static_assert(std::is_same<rm::LightBlob, rm::lightblob>::value && std::is_same<rm::Armour, rm::armour>::value,
              "north-star aliases (docs/core_8h_source.html:101,114)");
