from hipla.la import *       # noqa: F401,F403
from hipla.la import EigenValues_Preconditioner, InnerProduct  # noqa: F401
