/* units.c -- see units.h.  Clean-room stand-in for simutil's units.c. */
#include "units.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <ctype.h>

/* dimension vector order: length, mass, time, current, temperature */
#define NDIM 5
typedef struct { const char *name; double si; int dim[NDIM]; } unit_symbol;

static double internal_si[NDIM] = {1, 1, 1, 1, 1};
static double external_si[NDIM] = {1, 1, 1, 1, 1};
static int initialised = 0;
static char errmsg[256] = "";

const char *units_error(void) { return errmsg; }
void units_clear_error(void) { errmsg[0] = 0; }

void units_internal(double length, double mass, double time, double current, double temperature, double amount, double luminous)
{
   (void)amount; (void)luminous;
   internal_si[0] = length; internal_si[1] = mass; internal_si[2] = time; internal_si[3] = current; internal_si[4] = temperature;
   initialised = 1;
}
void units_external(double length, double mass, double time, double current, double temperature, double amount, double luminous)
{
   (void)amount; (void)luminous;
   external_si[0] = length; external_si[1] = mass; external_si[2] = time; external_si[3] = current; external_si[4] = temperature;
}
void units_ddcmd_defaults(void)
{
   /* ddcMD.c:71-72 */
   units_internal(a0_MKS, Rinfhc_MKS * 1e-30 / (a0_MKS * a0_MKS), 1e-15, e_MKS / 1e-15, Rinfhc_eV / kB_eV, 1.0, 1.0);
   units_external(1e-10, u_MKS, 1e-15, e_MKS / 1e-15, 1.0, 1.0, 1.0);
}

/* SI value and dimensions of the named units the decks use */
static const unit_symbol table[] = {
   {"Angstrom", 1e-10, {1, 0, 0, 0, 0}}, {"Ang", 1e-10, {1, 0, 0, 0, 0}}, {"nm", 1e-9, {1, 0, 0, 0, 0}},
   {"um", 1e-6, {1, 0, 0, 0, 0}}, {"bohr", a0_MKS, {1, 0, 0, 0, 0}}, {"a0", a0_MKS, {1, 0, 0, 0, 0}},
   {"amu", u_MKS, {0, 1, 0, 0, 0}}, {"M_p", mp_MKS, {0, 1, 0, 0, 0}}, {"M_e", me_MKS, {0, 1, 0, 0, 0}},
   {"g", 1e-3, {0, 1, 0, 0, 0}}, {"kg", 1.0, {0, 1, 0, 0, 0}},
   {"fs", 1e-15, {0, 0, 1, 0, 0}}, {"ps", 1e-12, {0, 0, 1, 0, 0}}, {"ns", 1e-9, {0, 0, 1, 0, 0}},
   {"us", 1e-6, {0, 0, 1, 0, 0}}, {"s", 1.0, {0, 0, 1, 0, 0}},
   {"e", e_MKS, {0, 0, 1, 1, 0}}, {"C", 1.0, {0, 0, 1, 1, 0}},
   {"K", 1.0, {0, 0, 0, 0, 1}},
   {"J", 1.0, {2, 1, -2, 0, 0}}, {"kJ", 1e3, {2, 1, -2, 0, 0}}, {"eV", e_MKS, {2, 1, -2, 0, 0}},
   {"meV", 1e-3 * e_MKS, {2, 1, -2, 0, 0}}, {"ueV", 1e-6 * e_MKS, {2, 1, -2, 0, 0}},      /* printinfo.c:36: the ENERGYFLUX default is ueV/Ang^2/fs */
   {"keV", 1e3 * e_MKS, {2, 1, -2, 0, 0}}, {"Ry", Rinfhc_MKS, {2, 1, -2, 0, 0}}, {"Rydberg", Rinfhc_MKS, {2, 1, -2, 0, 0}},
   {"Hartree", 2 * Rinfhc_MKS, {2, 1, -2, 0, 0}}, {"kcal", 4184.0, {2, 1, -2, 0, 0}},
   {"Pa", 1.0, {-1, 1, -2, 0, 0}}, {"bar", 1e5, {-1, 1, -2, 0, 0}}, {"GPa", 1e9, {-1, 1, -2, 0, 0}},
   {"Mbar", 1e11, {-1, 1, -2, 0, 0}}, {"atm", 101325.0, {-1, 1, -2, 0, 0}},
   {"mol", NA_MKS, {0, 0, 0, 0, 0}},     /* a pure number: kJ*mol^-1 is an energy per particle */
   {NULL, 0, {0, 0, 0, 0, 0}}};

/* evaluate a unit expression: SI factor and dimension vector */
static int eval_expr(const char *s, double *factor, int dim[NDIM])
{
   *factor = 1.0;
   for (int d = 0; d < NDIM; d++) dim[d] = 0;
   int sign = 1;   /* +1 after '*', -1 after '/' */
   const char *p = s;
   while (*p)
   {
      while (isspace((unsigned char)*p)) p++;
      if (!*p) break;
      if (*p == '*') { sign = 1; p++; continue; }
      if (*p == '/') { sign = -1; p++; continue; }
      double f = 1.0; int dm[NDIM] = {0, 0, 0, 0, 0};
      if (isdigit((unsigned char)*p) || *p == '.')
      {
         char *end; f = strtod(p, &end);
         /* a lone '.' ("kJ.mol^-1") is not a number: strtod consumes nothing */
         if (end == p) { snprintf(errmsg, sizeof(errmsg), "units: cannot parse '%s'", s); return -1; }
         p = end;
      }
      else if (isalpha((unsigned char)*p) || *p == '_')
      {
         char sym[64]; int n = 0;
         while ((isalnum((unsigned char)*p) || *p == '_') && n < 63) sym[n++] = *p++;
         sym[n] = 0;
         int found = 0;
         static const char *dimnames[NDIM] = {"l", "m", "t", "i", "T"};
         for (int d = 0; d < NDIM; d++)
            if (strcmp(sym, dimnames[d]) == 0) { f = external_si[d]; dm[d] = 1; found = 1; }
         if (!found)
            for (const unit_symbol *u = table; u->name; u++)
               if (strcmp(sym, u->name) == 0) { f = u->si; memcpy(dm, u->dim, sizeof(dm)); found = 1; break; }
         if (!found) { snprintf(errmsg, sizeof(errmsg), "units: unknown symbol '%s' in '%s'", sym, s); return -1; }
      }
      else { snprintf(errmsg, sizeof(errmsg), "units: cannot parse '%s'", s); return -1; }
      int power = 1;
      while (isspace((unsigned char)*p)) p++;
      if (*p == '^')
      {
         p++;
         char *end; power = (int)strtol(p, &end, 10);
         if (end == p) { snprintf(errmsg, sizeof(errmsg), "units: bad exponent in '%s'", s); return -1; }
         p = end;
      }
      power *= sign;
      *factor *= pow(f, power);
      for (int d = 0; d < NDIM; d++) dim[d] += power * dm[d];
      sign = 1;
   }
   return 0;
}

static double internal_factor(const int dim[NDIM])
{
   double f = 1.0;
   for (int d = 0; d < NDIM; d++) f *= pow(internal_si[d], dim[d]);
   return f;
}

double units_convert(double value, const char *from, const char *to)
{
   if (!initialised) units_ddcmd_defaults();
   double ff = 1.0, ft = 1.0;
   int df[NDIM], dt[NDIM];
   int have_from = (from != NULL), have_to = (to != NULL);
   if (have_from && eval_expr(from, &ff, df) != 0) return NAN;
   if (have_to && eval_expr(to, &ft, dt) != 0) return NAN;
   if (have_from && have_to)
   {
      for (int d = 0; d < NDIM; d++)
         if (df[d] != dt[d]) { snprintf(errmsg, sizeof(errmsg), "units: '%s' and '%s' differ in dimension", from, to); return NAN; }
      return value * ff / ft;
   }
   if (have_from) return value * ff / internal_factor(df);
   if (have_to) return value * internal_factor(dt) / ft;
   return value;
}

double units_ke(void)
{
   if (!initialised) units_ddcmd_defaults();
   /* e^2/(4 pi eps0) in J*m, expressed in internal energy*length (charge unit = e) */
   double ke_si = e_MKS * e_MKS / (4.0 * M_PI * eps0_MKS);
   int dim[NDIM] = {3, 1, -2, 0, 0};
   return ke_si / internal_factor(dim);
}
double units_kB(void)
{
   if (!initialised) units_ddcmd_defaults();
   int de[NDIM] = {2, 1, -2, 0, 0};
   return kB_MKS * internal_si[4] / internal_factor(de);
}
