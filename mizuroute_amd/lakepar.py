"""Order of the per-lake parameter rows `mzr_set_lakes` takes (include/mzr.h): the reference's names (popMetadat.f90:124-207,
dataTypes.f90:196-254), Doll 2003, HYPE, Hanasaki 2006 and its two memory switches."""
LAKE_PAR = ("D03_MaxStorage", "D03_Coefficient", "D03_Power", "D03_S0",
            "HYP_E_emr", "HYP_E_lim", "HYP_E_min", "HYP_E_zero", "HYP_Qrate_emr", "HYP_Erate_emr", "HYP_Qrate_prim",
            "HYP_Qrate_amp", "HYP_Qrate_phs", "HYP_prim_F", "HYP_A_avg", "HYP_Qsim_mode",
            "H06_Smax", "H06_alpha", "H06_envfact", "H06_S_ini", "H06_c1", "H06_c2", "H06_exponent", "H06_denominator",
            "H06_c_compare", "H06_frac_Sdead", "H06_E_rel_ini",
            "H06_I_Jan", "H06_I_Feb", "H06_I_Mar", "H06_I_Apr", "H06_I_May", "H06_I_Jun", "H06_I_Jul", "H06_I_Aug",
            "H06_I_Sep", "H06_I_Oct", "H06_I_Nov", "H06_I_Dec",
            "H06_D_Jan", "H06_D_Feb", "H06_D_Mar", "H06_D_Apr", "H06_D_May", "H06_D_Jun", "H06_D_Jul", "H06_D_Aug",
            "H06_D_Sep", "H06_D_Oct", "H06_D_Nov", "H06_D_Dec",
            "H06_purpose", "H06_I_mem_F", "H06_D_mem_F", "H06_I_mem_L", "H06_D_mem_L")
NLAKEPAR = len(LAKE_PAR)
