// compile-only: the C++ mirror must be self-contained (no HIP, no torch) and instantiate without the library present
#include <nrc_hpm.hpp>
int use(en::NeuralRadianceCache* c, en::NrcHpmRenderer* r, en::McHpmRenderer* m)
{
    return (c != nullptr) + (r != nullptr) + (m != nullptr) + (int)sizeof(en::AppConfig);
}
