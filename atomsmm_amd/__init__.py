"""atomsmm_amd -- MI355X-native drop-in for AtomsMM's RESPA-split nonbonded hot path.

    import atomsmm_amd as atomsmm
    from atomsmm_amd import openmm, unit            # stand-ins for `from simtk import openmm, unit`

Same class names and call signatures as `atomsmm` (reference v0.1.0) for the path BASELINE.json names:
DampedSmoothedForce / NearNonbondedForce / FarNonbondedForce / NonbondedExceptionsForce, RESPASystem,
RespaPropagator and friends.  The arithmetic runs in hand-written HIP kernels (libatomsmm_hip.so,
include/atomsmm_hip.h); importing this package does not need a GPU, creating a Context does.
"""
__version__ = '0.1.0'

from . import unit  # noqa: F401
from . import openmm  # noqa: F401
from .forces import DampedSmoothedForce  # noqa: F401
from .forces import FarNonbondedForce  # noqa: F401
from .forces import NearExceptionForce  # noqa: F401
from .forces import NearNonbondedForce  # noqa: F401
from .forces import NonbondedExceptionsForce  # noqa: F401
from .forces import SoftcoreForce  # noqa: F401
from .forces import SoftcoreLennardJonesForce  # noqa: F401
from .integrators import GlobalThermostatIntegrator  # noqa: F401
from .integrators import MultipleTimeScaleIntegrator  # noqa: F401
from .integrators import Langevin_R_Integrator, NHL_R_Integrator, SIN_R_Integrator  # noqa: F401
from .integrators import AdiabaticDynamicsIntegrator, ExtendedSystemVariable  # noqa: F401
from .propagators import ChainedPropagator  # noqa: F401
from .propagators import MultipleTimeScalePropagator  # noqa: F401
from .propagators import RespaPropagator  # noqa: F401
from .propagators import SplitPropagator  # noqa: F401
from .propagators import SuzukiYoshidaPropagator  # noqa: F401
from .propagators import TranslationPropagator  # noqa: F401
from .propagators import TrotterSuzukiPropagator  # noqa: F401
from .propagators import VelocityBoostPropagator  # noqa: F401
from .propagators import VelocityVerletPropagator  # noqa: F401
from .propagators import (MassiveNoseHooverPropagator, NoseHooverPropagator, OrnsteinUhlenbeckPropagator,  # noqa: F401
                          UnconstrainedVelocityVerletPropagator, VelocityRescalingPropagator,
                          GenericBoostPropagator, GenericScalingPropagator, MassiveIsokineticPropagator,
                          SIN_R_Propagator, MassiveGeneralizedGaussianMomentPropagator, NoseHooverChainPropagator,
                          NoseHooverLangevinPropagator)
from .systems import AlchemicalRespaSystem, AlchemicalSystem, ComputingSystem, RESPASystem, SolvationSystem  # noqa: F401
from .computers import PressureComputer  # noqa: F401
from .utils import InputError  # noqa: F401
from .utils import countDegreesOfFreedom  # noqa: F401
from .utils import evaluateForce  # noqa: F401
from .utils import findNonbondedForce  # noqa: F401
from .utils import hijackForce  # noqa: F401
from .utils import splitPotentialEnergy  # noqa: F401
from . import forces, integrators, propagators, systems, utils  # noqa: F401

__forces__ = ['DampedSmoothedForce', 'NonbondedExceptionsForce', 'NearExceptionForce', 'NearNonbondedForce',
              'FarNonbondedForce', 'SoftcoreLennardJonesForce', 'SoftcoreForce']
__integrators__ = ['GlobalThermostatIntegrator', 'MultipleTimeScaleIntegrator', 'Langevin_R_Integrator', 'NHL_R_Integrator', 'SIN_R_Integrator',
                   'AdiabaticDynamicsIntegrator', 'ExtendedSystemVariable']
__propagators__ = ['ChainedPropagator', 'MultipleTimeScalePropagator', 'RespaPropagator', 'SplitPropagator',
                   'SuzukiYoshidaPropagator', 'TranslationPropagator', 'TrotterSuzukiPropagator',
                   'VelocityBoostPropagator', 'VelocityVerletPropagator', 'UnconstrainedVelocityVerletPropagator',
                   'VelocityRescalingPropagator', 'NoseHooverPropagator', 'MassiveNoseHooverPropagator',
                   'OrnsteinUhlenbeckPropagator', 'GenericBoostPropagator', 'GenericScalingPropagator',
                   'MassiveIsokineticPropagator', 'SIN_R_Propagator', 'MassiveGeneralizedGaussianMomentPropagator',
                   'NoseHooverChainPropagator', 'NoseHooverLangevinPropagator']
__systems__ = ['RESPASystem', 'SolvationSystem', 'ComputingSystem', 'PressureComputer',
               'AlchemicalRespaSystem', 'AlchemicalSystem']
__utils__ = ['countDegreesOfFreedom', 'evaluateForce', 'findNonbondedForce', 'hijackForce', 'splitPotentialEnergy']
__all__ = __forces__ + __integrators__ + __propagators__ + __systems__ + __utils__
