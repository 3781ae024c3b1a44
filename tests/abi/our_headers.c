/* the same lines from the product's headers: host/plugin.h (INTEGRATOR) and include/ddcmi.h (label masks) */
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include "plugin.h"
#include "ddcmi.h"
#define OFF(T, m) printf(#T "." #m " %zu\n", offsetof(T, m))
int main(void)
{
   OFF(INTEGRATOR, name); OFF(INTEGRATOR, objclass); OFF(INTEGRATOR, value); OFF(INTEGRATOR, type); OFF(INTEGRATOR, parent); OFF(INTEGRATOR, itype);
   OFF(INTEGRATOR, uses_gpu); OFF(INTEGRATOR, eval_integrator); OFF(INTEGRATOR, writedynamic); OFF(INTEGRATOR, parms);
   printf("sizeof(INTEGRATOR) %zu\n", sizeof(INTEGRATOR));
   printf("NGLF %d\nNGLFCONSTRAINT %d\nNVTGLF %d\nHYCOPINTEGRATOR %d\n", (int)NGLF, (int)NGLFCONSTRAINT, (int)NVTGLF, (int)HYCOPINTEGRATOR);
   printf("sizeof(gid_type) %zu\n", sizeof(uint64_t));
   printf("molShift %d\n", DDCMI_GID_MOLSHIFT);
   printf("atmMask %016llx\natmgrpMask %016llx\ngrpMask %016llx\nresMask %016llx\nmolMask %016llx\nmolResMask %016llx\n",
          (unsigned long long)DDCMI_GID_ATMMASK, (unsigned long long)DDCMI_GID_ATMGRPMASK, (unsigned long long)DDCMI_GID_GRPMASK,
          (unsigned long long)DDCMI_GID_RESMASK, (unsigned long long)DDCMI_GID_MOLMASK, (unsigned long long)DDCMI_GID_MOLRESMASK);
   return 0;
}
