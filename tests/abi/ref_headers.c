/* Compiled by tests/test_abi.py ONLY where /root/reference exists, with -I/root/reference/src: the reference's OWN integrator.h,
 * bioGid.h and gid.h (three headers that compile by themselves) -- no transcription in between.  Prints the same "name value"
 * lines as our_headers.c prints from host/plugin.h and include/ddcmi.h. */
#include <stddef.h>
#include <stdio.h>
#include "integrator.h"
#include "bioGid.h"
#define OFF(T, m) printf(#T "." #m " %zu\n", offsetof(T, m))
int main(void)
{
   OFF(INTEGRATOR, name); OFF(INTEGRATOR, objclass); OFF(INTEGRATOR, value); OFF(INTEGRATOR, type); OFF(INTEGRATOR, parent); OFF(INTEGRATOR, itype);
   OFF(INTEGRATOR, uses_gpu); OFF(INTEGRATOR, eval_integrator); OFF(INTEGRATOR, writedynamic); OFF(INTEGRATOR, parms);
   printf("sizeof(INTEGRATOR) %zu\n", sizeof(INTEGRATOR));
   printf("NGLF %d\nNGLFCONSTRAINT %d\nNVTGLF %d\nHYCOPINTEGRATOR %d\n", (int)NGLF, (int)NGLFCONSTRAINT, (int)NVTGLF, (int)HYCOPINTEGRATOR);
   printf("sizeof(gid_type) %zu\n", sizeof(gid_type));
   printf("molShift %d\n", molShift);
   printf("atmMask %016llx\natmgrpMask %016llx\ngrpMask %016llx\nresMask %016llx\nmolMask %016llx\nmolResMask %016llx\n",
          (unsigned long long)atmMask, (unsigned long long)atmgrpMask, (unsigned long long)grpMask, (unsigned long long)resMask,
          (unsigned long long)molMask, (unsigned long long)molResMask);
   return 0;
}
