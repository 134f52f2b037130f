/* compile-only: the C ABI header must be plain C */
#include <nrc_hpm.h>
int use(const nrc_config* c, const nrc_scene* s, const nrc_camera* k, const nrc_tile* t) { return c && s && k && t; }
