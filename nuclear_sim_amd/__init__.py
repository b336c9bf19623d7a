"""MI355X-native batched plant-dynamics stepper (drop-in for nuclear-sim's step()/reset() path)."""
